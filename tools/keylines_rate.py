#!/usr/bin/env python3
"""Rate of the EDLines / multi-octave KeyLine path (lf_keylines_batch): a 256-frame batch of the bench's
synthetic 640x480 lane frames, full-res geometry, EDLines over N octaves + LBD on the detector's gradients.  Frames are
resident in HBM; per-stage times by HIP events on the handle's stream (the sequential smart routing + line fitting is
accounted under the stage LSD's region growing uses).

    python tools/keylines_rate.py [--octaves 1,3] [--frames 256] [--reps 5] [--content synthetic|clutter|real]

--content clutter: the same frames + speckle and 40 random strokes in lane colours each (bench.py's clutter); real: the three
Duckiebot camera frames of tests/golden/real_frames.npz tiled to the batch, each copy shifted by 7 px.
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
ap.add_argument("--content", default="synthetic", choices=("synthetic", "clutter", "real"))
args = ap.parse_args()
torch.cuda.init()
dev = torch.device("cuda")
cfg = default_config(args.geometry)
B = args.frames
frames = synth.make_batch(B, seed0=0, threads=max(1, min(32, (os.cpu_count() or 2) // 2)))
if args.content == "clutter":
    rng = np.random.default_rng(4321)
    for f in range(B):
        img = frames[f]
        r0 = img.shape[0] // 3
        for _ in range(40):
            y, x = rng.integers(r0 + 10, img.shape[0] - 10), rng.integers(10, img.shape[1] - 10)
            dy, dx = rng.integers(-12, 13), rng.integers(-40, 41)
            col = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[rng.integers(0, 3)]
            t = np.linspace(0, 1, 80)
            yy = np.clip((y + t * dy + rng.normal(0, 0.7, 80)).astype(int), r0, img.shape[0] - 1)
            xx = np.clip((x + t * dx + rng.normal(0, 0.7, 80)).astype(int), 0, img.shape[1] - 1)
            img[yy, xx] = col
        img[rng.random(img.shape[:2]) < 0.004] = (235, 235, 235)
elif args.content == "real":
    real = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_frames.npz"))
    rf = [real[k] for k in real.files if real[k].ndim == 3 and real[k].shape == frames.shape[1:]]
    frames = np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(B)])
print("content: %s" % args.content)
d_frames = torch.from_numpy(frames).to(dev)
fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=256)
cap = B * 4096
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
    cnt = fe.keylines_fetch(0, 6, B)
    sid = fe.keylines_fetch(0, 5, B)
    chain_px = [int(sid[f][cnt[f, 1]]) if cnt[f, 1] >= 0 else 0 for f in range(B)]
    print("    octave 0 per frame: %.0f anchors, %.1f chains, %.0f chain pixels, %.1f lines (max %d anchors, %d chains, %d chain pixels)"
          % (cnt[:, 0].mean(), cnt[:, 1].mean(), np.mean(chain_px), cnt[:, 2].mean(), cnt[:, 0].max(), cnt[:, 1].max(), max(chain_px)))
fe.close()
