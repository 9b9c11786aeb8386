#!/usr/bin/env python3
"""Associator stress (BASELINE configs[4]): exact Hamming nearest neighbour of N 256-bit query codes in an M-code
live map (lane_slam_amd.LineAssociator: the map's int8 operands stay packed on the device), int8 MFMA path; reports the
per-call time of the association (HIP events on the map's stream) and 2*N*M*256 ops/s against the dense int8 peak.

    python tools/assoc_rate.py [--pairs 4096x50000,16384x50000,65536x50000,65536x262144] [--gating]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # kernel arguments in device memory: ~2 us less per launch
import torch
from lane_slam_amd import LineAssociator, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", default="4096x50000,16384x50000,65536x50000,65536x262144")
ap.add_argument("--gating", action="store_true", help="colour-gated association (same kernel, colour terms in the ninth MFMA step)")
ap.add_argument("--tie-rule", default="lowest", choices=("lowest", "mihasher"), help="mihasher: + the tie-ranking pass (k_assoc_ties.hip)")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
torch.cuda.init()
PEAK = 5.0e15          # dense int8 MFMA peak (ops/s); the FP4 peak is twice that
int8_forced = os.environ.get("LF_ASSOC_INT8") is not None
for pair in args.pairs.split(","):
    n, m = (int(v) for v in pair.split("x"))
    am = LineAssociator(capacity=max(64, m), color_gating=args.gating, kept_only=False, tie_rule=args.tie_rule)
    rng = np.random.default_rng(3)
    am.seed(synth.random_codes(m, 2), rng.integers(0, 3, m).astype(np.uint8))
    q = torch.from_numpy(synth.random_codes(n, 1)).cuda()
    qc = torch.from_numpy(rng.integers(0, 3, n).astype(np.uint8)).cuda()
    idx = torch.zeros(n, dtype=torch.int32, device="cuda")
    dist = torch.zeros(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        am.associate_device(None, q.data_ptr(), qc.data_ptr(), n, idx.data_ptr(), dist.data_ptr())
    am.synchronize()
    am.timing()
    am.set_profiling(True)
    for _ in range(args.reps):
        am.associate_device(None, q.data_ptr(), qc.data_ptr(), n, idx.data_ptr(), dist.data_ptr())
    am.synchronize()
    t = am.timing()
    core = t["assoc_mfma"][0] / t["assoc_mfma"][1]
    if t["assoc_pack_queries"][1] > 0:        # libraries before the single-launch associator packed the queries in a kernel of their own
        core += t["assoc_pack_queries"][0] / t["assoc_pack_queries"][1]
    ops = 2.0 * n * m * 256
    fp4 = not int8_forced          # v_mfma_scale_f32_32x32x64_f8f6f4 with e2m1 operands (LF_ASSOC_INT8=1: the int8 kernels, for A/B)
    rate = ops / (core * 1e-3)
    print("N=%6d M=%7d%s%s: assoc %.4f ms (one launch: query expansion, MFMA, merge, report)  %.2f Pop/s  %s"
          % (n, m, " gated" if args.gating else "", " tie_rule=mihasher (two passes)" if args.tie_rule == "mihasher" else "", core, rate / 1e15,
             ("FP4 kernel: %.1f %% of the 10 Pop/s dense FP4 peak (= %.1f %% of the 5 Pop/s int8 peak the int8 kernel is priced against)" % (100 * rate / (2 * PEAK), 100 * rate / PEAK))
             if fp4 else ("int8 kernel: %.1f %% of the 5 Pop/s dense int8 MFMA peak" % (100 * rate / PEAK))))
    am.close()
