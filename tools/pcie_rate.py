#!/usr/bin/env python3
"""PCIe-inclusive throughput: frames start in host memory (pinned via torch if available) and every
step pays the H2D copy of its 256 frames; outputs stay on the device.  Reported in DESIGN.md next to
the HBM-resident headline number (never used as bench.py's `value`)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lane_slam_amd import FrontEnd, default_config, synth

B, D, steps = 256, 3, 18
cfg = default_config("fullres")
fes = [FrontEnd(cfg, max_frames=B, max_lines_per_color=512) for _ in range(D)]
host = np.ascontiguousarray(np.tile(synth.make_batch(32, 0), (8, 1, 1, 1)))
pinned = torch.from_numpy(host).pin_memory()
cap = B * 3 * 512
dev = torch.device("cuda", 0)
outs = [{k: torch.zeros(*shp, dtype=dt, device=dev) for k, shp, dt in (
    ("frame_offset", (B + 1,), torch.int32), ("lines", (cap, 4), torch.float32), ("normals", (cap, 2), torch.float32),
    ("color", (cap,), torch.uint8), ("pixels_normalized", (cap, 4), torch.float32), ("ground", (cap, 4), torch.float64),
    ("keep", (cap,), torch.uint8), ("desc", (cap, 72), torch.float32), ("code", (cap, 32), torch.uint8))} for _ in range(D)]
ptrs = [{k: v.data_ptr() for k, v in o.items()} for o in outs]
import ctypes
from lane_slam_amd import _lib

def submit_host(fe, p):
    s = _lib.LfSegments(); s.capacity = cap
    for k, v in p.items(): setattr(s, k, int(v))
    fe._check(fe.lib.lf_process_batch_async(fe.h, ctypes.c_void_p(pinned.data_ptr()), B, 0, ctypes.byref(s), 1))

def run(n):
    infl = []
    for k in range(n):
        sl = k % D
        if len(infl) == D: fes[infl.pop(0)].wait()
        submit_host(fes[sl], ptrs[sl]); infl.append(sl)
    while infl: fes[infl.pop(0)].wait()

run(4); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("PCIe-inclusive (pinned host frames, H2D every step, no association): %.0f frames/s, %.2f ms/step, H2D %.1f GB/s of payload"
      % (B * steps / dt, 1e3 * dt / steps, B * steps * 480 * 640 * 3 / dt / 1e9))
