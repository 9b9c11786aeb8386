#!/usr/bin/env python3
"""Time of one live-map update (lf_map_update: k_map_classify + k_map_plan + k_map_apply) with the block shapes of an
N-rank step (BASELINE configs[3]): n_blocks gathered blocks of 16 Ki rows each, about 11 k segments per
block -- what EVERY rank runs after the all-gather, every step.  HIP events on the map's stream.

    python tools/map_update_rate.py [--blocks 1,2,4,8] [--rows 16384] [--segments 11103]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from lane_slam_amd import LineAssociator, synth
from lane_slam_amd.distributed import BLOCK_ROW_BYTES

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", default="1,2,4,8")
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--segments", type=int, default=11103)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--policy", default="append")
args = ap.parse_args()
torch.cuda.init()
dev = torch.device("cuda")
rows = args.rows + 1
for nb in (int(v) for v in args.blocks.split(",")):
    am = LineAssociator(capacity=66384, color_gating=False, kept_only=True, policy=args.policy, merge_distance=20 if args.policy == "merge" else 0)
    rng = np.random.default_rng(nb)
    am.seed(synth.random_codes(66384, 2))
    blocks = torch.zeros(nb * rows * BLOCK_ROW_BYTES, dtype=torch.uint8, device=dev)
    n = args.segments
    keep = []
    for b in range(nb):
        out = {"frame_offset": torch.tensor([0, n], dtype=torch.int32, device=dev), "code": torch.from_numpy(synth.random_codes(n, 10 + b)).to(dev),
               "color": torch.from_numpy(rng.integers(0, 3, n).astype(np.uint8)).to(dev), "keep": torch.from_numpy((rng.random(n) < 0.5).astype(np.uint8)).to(dev),
               "ground": torch.from_numpy(rng.normal(size=(n, 4))).to(dev)}
        idx = torch.zeros(n, dtype=torch.int32, device=dev)
        dd = torch.zeros(n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ptrs = {k: v.data_ptr() for k, v in out.items()}
        am.associate_device(None, ptrs["code"], ptrs["color"], n, idx.data_ptr(), dd.data_ptr())
        am.pack_block_device(None, ptrs, n, 1, idx.data_ptr(), dd.data_ptr(), None, 0, blocks.data_ptr() + b * rows * BLOCK_ROW_BYTES, rows)
        keep.append((out, idx, dd))
    am.synchronize()
    for _ in range(3):
        am.update_device(blocks.data_ptr(), nb, rows)
    am.synchronize()
    am.timing()
    am.set_profiling(True)
    for _ in range(args.reps):
        am.update_device(blocks.data_ptr(), nb, rows)
    am.synchronize()
    t = am.timing()["map_update"]
    print("n_blocks=%d x %d rows (%d segments each, %s): map update %.4f ms per call (3 launches: classify, plan, apply); map %r"
          % (nb, args.rows, n, args.policy, t[0] / t[1], am.state()))
    am.close()
