#!/usr/bin/env python3
"""Rate of the EDLines / multi-octave KeyLine path (lf_keylines_batch): a 256-frame batch of the bench's
synthetic 640x480 lane frames, full-res geometry, EDLines over N octaves + LBD on the detector's gradients.  Frames are
resident in HBM; per-stage times by HIP events on the handle's stream (the sequential smart routing + line fitting is
accounted under the stage LSD's region growing uses).

    python tools/keylines_rate.py [--octaves 1,3] [--frames 256] [--reps 5] [--clutter]
"""
import argparse, ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from lane_slam_amd import FrontEnd, default_config, synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--octaves", default="1,3")
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--geometry", default="fullres")
args = ap.parse_args()
torch.cuda.init()
dev = torch.device("cuda")
cfg = default_config(args.geometry)
B = args.frames
frames = synth.make_batch(B, seed0=0, threads=max(1, min(32, (os.cpu_count() or 2) // 2)))
d_frames = torch.from_numpy(frames).to(dev)
fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=256)
cap = B * 512
out = {k: torch.zeros((cap, c) if c > 1 else cap, dtype={"f4": torch.float32, "i4": torch.int32, "u1": torch.uint8}[dt], device=dev) for k, dt, c in _lib.KEYLINE_FIELDS}
fo = torch.zeros(B + 1, dtype=torch.int32, device=dev)
for n_oct in (int(v) for v in args.octaves.split(",")):
    s = _lib.LfKeylines()
    s.capacity = cap
    s.frame_offset = fo.data_ptr()
    for k, _, _ in _lib.KEYLINE_FIELDS:
        setattr(s, k, out[k].data_ptr())
    total = ctypes.c_int()

    def run():
        fe._check(fe.lib.lf_keylines_batch(fe.h, ctypes.c_void_p(d_frames.data_ptr()), B, 0, 1, n_oct, None, ctypes.byref(s), 1, 1, ctypes.byref(total), None))
    torch.cuda.synchronize()
    run(); run()
    fe.set_profiling(True); fe.reset_timing()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    dt = (time.perf_counter() - t0) / args.reps
    t = fe.timing()
    fe.set_profiling(False)
    stages = ", ".join("%s %.3f ms" % (k.split("(")[0], v[0] / args.reps) for k, v in t.items() if v[1])
    print("octaves=%d: %d KeyLines per %d-frame batch (%.1f per frame), %.3f ms per batch = %.1f k frames/s (synchronous call); %s"
          % (n_oct, total.value, B, total.value / B, dt * 1e3, B / dt / 1e3, stages))
fe.close()
