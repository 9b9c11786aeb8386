#!/usr/bin/env python3
"""Diagnostic: per-phase cycle totals of k_lsd_grow.  usage: grow_stamps.py [frames] [geometry].  Needs a library built with
`make -C lane_slam_amd/csrc EXTRA=-DLFG_STAMPS` (never the shipped build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_slam_amd import FrontEnd, default_config, synth
from lane_slam_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = default_config(sys.argv[2] if len(sys.argv) > 2 else "fullres")      # "parity": the 160x120 operating point of the reference
fe = FrontEnd(cfg, max_frames=n, max_lines_per_color=512)
frames = synth.make_batch(n, 0)
if os.environ.get("LF_STAMPS_REAL"):          # the three real camera frames of tests/golden, tiled as bench.py's secondary.real_frames does
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_jpegs.npz"))
    streams = [bytes(z["jpeg%02d" % k]) for k in range(len(z["names"]))]
    fr, st = fe.decode_jpeg_batch(streams, n_threads=4)
    rf = [fr[k] for k in range(len(streams)) if st[k] == 0]
    frames = np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(n)])
if os.environ.get("LF_STAMPS_CLUTTER"):       # bench.py's secondary.clutter_frames: speckle + 40 random strokes in lane colours per frame
    rng_c = np.random.default_rng(4321)
    for f_ in range(frames.shape[0]):
        img = frames[f_]
        r0 = img.shape[0] // 3
        for _ in range(40):
            y, x = rng_c.integers(r0 + 10, img.shape[0] - 10), rng_c.integers(10, img.shape[1] - 10)
            dy, dx = rng_c.integers(-12, 13), rng_c.integers(-40, 41)
            col = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[rng_c.integers(0, 3)]
            t_ = np.linspace(0, 1, 80)
            yy = np.clip((y + t_ * dy + rng_c.normal(0, 0.7, 80)).astype(int), r0, img.shape[0] - 1)
            xx = np.clip((x + t_ * dx + rng_c.normal(0, 0.7, 80)).astype(int), 0, img.shape[1] - 1)
            img[yy, xx] = col
        img[rng_c.random(img.shape[:2]) < 0.004] = (235, 235, 235)
fe.process_batch(frames)
seg = fe.process_batch(frames)
scr = fe.fetch(_lib.LF_BUF_LSD_SCRATCH, n)
Ps = scr.shape[2]
# GROW_WAVES (4: three growing + the evaluating one) waves per problem, 32 u64 each, in the last 64 * GROW_WAVES words of the problem's region scratch
GW = int(os.environ.get("LFG_GW", "4"))
raw = scr[:, :, Ps - 64 * GW:].copy().view(np.uint64).reshape(n * 3, GW, 32)
if os.environ.get("LFG_EVAL_KERNEL") == "1":      # library built with -DLFG_EVAL_KERNEL=1 -DLFG_STAMPS (and LFG_GW=3)
    EW = int(os.environ.get("LFG_EW", "4"))
    ev = scr[:, :, Ps - 64 * GW - 64 * EW:Ps - 64 * GW].copy().view(np.uint64).reshape(n * 3, EW, 32)
    et = ev[:, :, 24].astype(np.float64)
    print("k_lsd_eval waves (kcycles): slowest wave of a problem mean %.0f max %.0f | sum over waves mean %.0f | pending regions mean %.1f max %d" % (
        et.max(1).mean() / 1e3, et.max() / 1e3, et.sum(1).mean() / 1e3, ev[:, 0, 25].mean(), ev[:, 0, 25].max()))
    for w in np.argsort(et.max(1))[-4:]:
        k = et[w].argmax()
        print("  longest evaluating wave: problem %d, %d pending, total %.0f scan %.0f math %.0f kcycles, nfa calls %d px %d" % (
            w, ev[w, 0, 25], et[w, k] / 1e3, ev[w, k, 4] / 1e3, ev[w, k, 5] / 1e3, ev[w, k, 7] >> 40, ev[w, k, 7] & ((1 << 40) - 1)))
tot_w = raw[:, :, 24].astype(np.float64)
print("wave totals (kcycles): slowest wave mean %.0f max %.0f | sum over waves mean %.0f | components mean %.1f" % (
    tot_w.max(1).mean() / 1e3, tot_w.max() / 1e3, tot_w.sum(1).mean() / 1e3, raw[:, 0, 26].mean()))
slow = tot_w.argmax(1)
worst = np.argsort(tot_w.max(1))[-4:]
print("waves of the four longest problems (kcycles):", [[int(v / 1e3) for v in tot_w[w]] for w in worst])
if GW == 4:
    for w in worst:
        print("  problem %d evaluating wave: busy %.0f kcycles (scan %.0f + math %.0f) of %.0f, %d nfa calls" % (
            w, (raw[w, 3, 4] + raw[w, 3, 5]) / 1e3, raw[w, 3, 4] / 1e3, raw[w, 3, 5] / 1e3, tot_w[w, 3] / 1e3, raw[w, 3, 7] >> 40))
d = raw[np.arange(n * 3), slow][:, :32].astype(np.float64)          # the slowest wave of every problem
names = ["seed", "grow", "rect", "refine", "nfa_scan", "nfa_math", "improve+emit", "-", "fetch", "regions", "reg_pts", "batches",
         "g_lookup", "g_fallback", "g_accept", "fallback_batches", "g_window", "refine_rect", "bulk_acc", "exact_acc", "isolated_seeds", "refine_tau", "refine_regrow", "refine_reduce"] + ["total", "n_order", "n_comp", "spans", "spans_pure_undecided", "undecided_rejected", "spans_dupwalk", "push"]
CYC = {0, 1, 2, 3, 4, 5, 6, 8, 12, 13, 14, 16, 17, 21, 22, 23, 24, 31}
raw7 = raw[:, :, 7].sum(1)
print("nfa calls mean/max", (raw7 >> 40).mean(), (raw7 >> 40).max(), "px tested mean/max", (raw7 & ((1 << 40) - 1)).mean(), (raw7 & ((1 << 40) - 1)).max())
# the growing waves (0..2) alone: the evaluating wave's total is "until the last grower is done" and hides them
gtot = tot_w[:, :min(3, GW)]
gslow = gtot.argmax(1)
dg = raw[np.arange(n * 3), gslow][:, :32].astype(np.float64)
for w in np.argsort(gtot.max(1))[-3:]:
    print("longest growers: problem", int(w), " ".join("%s=%.0f" % (nm, v / (1000.0 if i in CYC else 1.0)) for i, (nm, v) in enumerate(zip(names, dg[w])) if nm != "-"), "(kcycles)")
print("growers mean", " ".join("%s=%.0f" % (nm, v / (1000.0 if i in CYC else 1.0)) for i, (nm, v) in enumerate(zip(names, dg.mean(0))) if nm != "-"), "(kcycles)")
order = np.argsort(d[:, 24])
print("problems", n * 3, "segments", seg.n)
for label, rows in (("median", d[order[len(order) // 2]]), ("p90", d[order[int(len(order) * 0.9)]]), ("max", d[order[-1]]), ("mean", d.mean(0))):
    print(label, " ".join("%s=%.0f" % (nm, v / (1000.0 if i in CYC else 1.0)) for i, (nm, v) in enumerate(zip(names, rows)) if nm != "-"), "(kcycles)")
