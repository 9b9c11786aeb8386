#!/usr/bin/env python3
"""Per-stage kernel times (one batch in flight) of the front end on other content than the bench's synthetic lane frames:
the three real camera frames tiled to a batch (bench.py's secondary.real_frames) and the clutter frames.
    python tools/content_profile.py [--batch 256] [--steps 6]"""
import argparse, os, sys
import numpy as np
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_slam_amd import FrontEnd, default_config, synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--only", default="", help="one of synthetic / clutter / real (for a run under rocprofv3 --kernel-trace --stats)")
args = ap.parse_args()
B = args.batch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = default_config("fullres")
def real_sample():
    """the 28 camera frames of tests/golden/real_jpegs.npz, decoded by the device decoder"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "real_jpegs.npz"))
    streams = [bytes(z["jpeg%02d" % k]) for k in range(len(z["names"]))]
    fe0 = FrontEnd(cfg, max_frames=len(streams), max_lines_per_color=64)
    fr, st = fe0.decode_jpeg_batch(streams, n_threads=4)
    fe0.close()
    assert (st == 0).all()
    return [fr[k] for k in range(len(streams))]
rf = real_sample()
def clutter(frames, seed):
    rng = np.random.default_rng(seed)
    out = frames.copy()
    for img in out:
        r0 = img.shape[0] // 3
        for _ in range(40):
            y, x = rng.integers(r0 + 10, img.shape[0] - 10), rng.integers(10, img.shape[1] - 10)
            dy, dx = rng.integers(-12, 13), rng.integers(-40, 41)
            col = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[rng.integers(0, 3)]
            t_ = np.linspace(0, 1, 80)
            yy = np.clip((y + t_ * dy + rng.normal(0, 0.7, 80)).astype(int), r0, img.shape[0] - 1)
            xx = np.clip((x + t_ * dx + rng.normal(0, 0.7, 80)).astype(int), 0, img.shape[1] - 1)
            img[yy, xx] = col
        m = rng.random(img.shape[:2]) < 0.004
        m[:r0] = False
        img[m] = (235, 235, 235)
    return out


workloads = {"synthetic": synth.make_batch(B, 0), "clutter": clutter(synth.make_batch(B, 0), 4321), "real": np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(B)])}
for name, frames in workloads.items():
    if args.only and name != args.only:
        continue
    fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=512)
    fe.set_profiling(True)
    seg = None
    for _ in range(3):
        seg = fe.process_batch(frames, describe=True)
    fe.reset_timing()                                      # drop what the warm-up steps measured
    for _ in range(args.steps):
        seg = fe.process_batch(frames, describe=True)
    t = fe.timing()
    tot = sum(ms / max(l, 1) for ms, l in t.values() if l)
    print("%s: %d segments per frame, %.3f ms of kernels per %d-frame batch" % (name, seg.n // B, tot, B))
    nd = fe.fetch(_lib.LF_BUF_LSD_NORDER, B) if hasattr(_lib, "LF_BUF_LSD_NORDER") else None
    if nd is not None:
        print("   defined pixels per problem: mean %.0f, p90 %.0f, max %d" % (nd.mean(), np.percentile(nd, 90), nd.max()))
    for k, (ms, l) in t.items():
        if l:
            print("   %-40s %.4f ms" % (k, ms / l))
    fe.close()
