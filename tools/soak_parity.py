#!/usr/bin/env python3
"""Soak test: many synthetic frames through the whole GPU path against the oracle, bit for bit.
Not part of the pytest suites (minutes of single-threaded oracle time); run on the GPU box:

    python tools/soak_parity.py [--parity 2000] [--fullres 200] [--clutter 100]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_slam_amd import FrontEnd, default_config, synth
from oracle.oracle import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--parity", type=int, default=2000)
ap.add_argument("--fullres", type=int, default=200)
ap.add_argument("--clutter", type=int, default=100)
ap.add_argument("--seed", type=int, default=0, help="offset added to every frame seed")
ap.add_argument("--hd", type=int, default=0, help="1920x1080 frames (img_size [1080,1920], top_cutoff 360)")
args = ap.parse_args()
KEYS = ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "code")


def check(geom, n, seed0, mutate=None):
    hd = geom == "hd"
    cfg = default_config("fullres", in_size=(1080, 1920)) if hd else default_config(geom)
    o = Oracle(cfg)
    B = 16 if hd else 64
    fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=4096)
    bad = segs = 0
    t0 = time.time()
    for b0 in range(0, n, B):
        nb = min(B, n - b0)
        frames = synth.make_batch(nb, seed0 + b0, rows=1080, cols=1920) if hd else synth.make_batch(nb, seed0 + b0)
        if mutate is not None:
            frames = mutate(frames, seed0 + b0)
        seg = fe.process_batch(frames)
        for f in range(nb):
            r = o.process_frame(frames[f], cap=16384)
            s = seg.frame(f)
            ok = s.n == r["n"] and all(np.array_equal(getattr(s, k), r[k]) for k in KEYS)
            ok = ok and np.allclose(s.desc, r["desc"], rtol=0, atol=1e-6)
            segs += r["n"]
            if not ok:
                bad += 1
                print("MISMATCH", geom, "seed", seed0 + b0 + f, "n", s.n, r["n"], flush=True)
    print("%s: %d frames, %d segments, %d mismatching frames, %.0f s" % (geom, n, segs, bad, time.time() - t0), flush=True)
    return bad


def clutter(frames, seed):
    """Speckle + random bright strokes in lane colours: many short, noisy regions (refine / rect_improve / NFA rejections)."""
    rng = np.random.default_rng(seed)
    out = frames.copy()
    for f in range(out.shape[0]):
        img = out[f]
        for _ in range(40):
            y, x = rng.integers(170, 470), rng.integers(10, 630)
            dy, dx = rng.integers(-12, 13), rng.integers(-40, 41)
            col = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[rng.integers(0, 3)]
            for s in np.linspace(0, 1, 80):
                yy, xx = int(y + s * dy + rng.normal(0, 0.7)), int(x + s * dx + rng.normal(0, 0.7))
                if 160 <= yy < 480 and 0 <= xx < 640:
                    img[yy, xx] = col
        m = rng.random(img.shape[:2]) < 0.004
        img[m] = (235, 235, 235)
    return out


bad = 0
bad += check("parity", args.parity, 100000 + args.seed)
bad += check("fullres", args.fullres, 200000 + args.seed)
bad += check("fullres", args.clutter, 300000 + args.seed, clutter)
if args.hd:
    bad += check("hd", args.hd, 400000 + args.seed)
print("soak:", "OK" if bad == 0 else "%d MISMATCHING FRAMES" % bad)
sys.exit(1 if bad else 0)
