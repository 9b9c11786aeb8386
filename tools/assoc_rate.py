#!/usr/bin/env python3
"""Associator stress (BASELINE configs[4]): exact Hamming nearest neighbour of N 256-bit query codes in an M-code
map on the int8 MFMA path, device resident; reports time and 2*N*M*256 ops/s against the dense int8 peak.

    python tools/assoc_rate.py [--pairs 4096x50000,16384x50000,65536x50000,65536x262144]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lane_slam_amd import FrontEnd, default_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", default="4096x50000,16384x50000,65536x50000,65536x262144")
args = ap.parse_args()
torch.cuda.init()
fe = FrontEnd(default_config("parity"), max_frames=1, max_lines_per_color=16)
PEAK = 5.0e15
for pair in args.pairs.split(","):
    n, m = (int(v) for v in pair.split("x"))
    q = torch.from_numpy(synth.random_codes(n, 1)).cuda()
    mp = torch.from_numpy(synth.random_codes(m, 2)).cuda()
    idx = torch.zeros(n, dtype=torch.int32, device="cuda")
    dist = torch.zeros(n, dtype=torch.float32, device="cuda")
    for _ in range(3):
        fe.associate_device(q.data_ptr(), n, mp.data_ptr(), m, idx.data_ptr(), dist.data_ptr())
    fe.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        fe.associate_device(q.data_ptr(), n, mp.data_ptr(), m, idx.data_ptr(), dist.data_ptr())
    fe.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ops = 2.0 * n * m * 256
    print("N=%6d M=%7d: %.3f ms  %.2f Pop/s  (%.0f %% of the dense int8 MFMA peak), incl. the +-64 packing of both sides" % (n, m, dt * 1e3, ops / dt / 1e15, 100 * ops / dt / PEAK))
