#!/usr/bin/env python3
"""Soak test of the EDLines / multi-octave KeyLines / LBD path (lf_keylines_batch) against the oracle, bit for bit: synthetic
lane frames, the same with clutter (speckle + strokes), and random noise-plus-shapes images, 1..4 octaves, default and
modified parameters.  Not part of the pytest suites; run on the GPU box:

    python tools/soak_edlines.py [--frames 400] [--seed 0]
"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_slam_amd import FrontEnd, default_config, synth
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=400)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
FIELDS = ("start_end", "in_octave", "angle", "num_pixels", "line_length", "octave", "class_id", "response", "size", "pt", "code")


def clutter(frames, seed):
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


def shapes(n, rows, cols, seed):
    """smooth random polygons and bars on a noisy gradient background (supersampled: EDLines needs soft edges)"""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, rows, cols), np.uint8)
    S = 3
    yy, xx = np.mgrid[0:rows * S, 0:cols * S]
    for f in range(n):
        big = 90.0 + 40.0 * np.sin(xx / (cols * S) * rng.uniform(1, 5) + rng.uniform(0, 6)) * np.cos(yy / (rows * S) * rng.uniform(1, 4))
        for _ in range(int(rng.integers(4, 14))):
            cx, cy = rng.uniform(0, cols * S), rng.uniform(0, rows * S)
            a = rng.uniform(0, np.pi)
            u = (xx - cx) * np.cos(a) + (yy - cy) * np.sin(a)
            v = -(xx - cx) * np.sin(a) + (yy - cy) * np.cos(a)
            inside = (np.abs(u) < rng.uniform(20, 300)) & (np.abs(v) < rng.uniform(4, 60))
            big[inside] = rng.uniform(10, 245)
        small = big.reshape(rows, S, cols, S).mean(axis=(1, 3)) + rng.normal(0, rng.uniform(0, 4), (rows, cols))
        out[f] = np.clip(np.rint(small), 0, 255).astype(np.uint8)
    return out


def check(name, gray, n_octaves, params_kw, fe):
    params = fe.edlines_params(**params_kw) if params_kw else None
    k = fe.keylines_batch(gray, n_octaves=n_octaves, gray=True, params=params, capacity=gray.shape[0] * 8000)
    bad = lines = 0
    for f in range(gray.shape[0]):
        r = O.octave_keylines(gray[f], n_octaves, O.edlines_params(**params_kw) if params_kw else None, cap=40000)
        a, b = int(k["frame_offset"][f]), int(k["frame_offset"][f + 1])
        st = int(k["frame_status"][f])
        if r is None or st != 0:
            ok = (r is None) == (st != 0) and b == a              # the detector gave up: on both sides, and no KeyLines
        else:
            ok = b - a == r["n"] and all(np.array_equal(k[name_][a:b], r[name_]) for name_ in FIELDS)
            ok = ok and np.array_equal(k["desc"][a:b], r["desc"], equal_nan=True)
            lines += r["n"]
        if not ok:
            bad += 1
            print("MISMATCH", name, "frame", f, "octaves", n_octaves, params_kw, b - a, None if r is None else r["n"], st, flush=True)
    return bad, lines


cfg = default_config("fullres")
o = O.Oracle(cfg)
B = 8
fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=256)
bad = total = frames_done = 0
t0 = time.time()
variants = [(1, {}), (3, {}), (2, {"gradient_threshold": 40, "anchor_threshold": 4}), (4, {"min_line_len": 25, "line_fit_err_threshold": 2.2}),
            (3, {"scan_intervals": 1, "gradient_threshold": 120})]
b0 = 0
while frames_done < args.frames:
    seed = 500000 + args.seed + b0
    kind = (b0 // B) % 3
    if kind == 2:
        gray = shapes(B, cfg["img_size"][0] - cfg["top_cutoff"], cfg["img_size"][1], seed)
        name = "shapes"
    else:
        fr = synth.make_batch(B, seed)
        if kind == 1:
            fr = clutter(fr, seed)
        gray = np.stack([o.bgr2gray(o.preprocess(f)) for f in fr])
        name = "clutter" if kind == 1 else "lane"
    n_oct, kw = variants[(b0 // B) % len(variants)]
    bb, ll = check(name, gray, n_oct, kw, fe)
    bad += bb; total += ll; frames_done += B; b0 += B
print("edlines soak: %d frames, %d KeyLines, %d mismatching frames, %.0f s" % (frames_done, total, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
