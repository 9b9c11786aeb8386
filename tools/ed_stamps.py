#!/usr/bin/env python3
"""Where the EDLines walking wave spends its cycles: runs lf_keylines_batch (one octave) on the bench's frames with every diagnostic
build lane_slam_amd/liblanefront_ed<k>.so (k_edlines.hip, -DLF_ED_STAMP=k; built by tools/ed_stamps.sh) and prints the per-frame mean and
the slowest frame's value of each quantity.

    python tools/ed_stamps.py [--content synthetic|clutter|real]
"""
import argparse, ctypes, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {1: "cycles inside ed_walk", 2: "  a window left -> the walk goes on", 3: "  a walk's start", 4: "windows fetched on leaving one", 5: "windows fetched at a start",
         6: "steps", 7: "cycles kernel start -> walk begins", 8: "cycles walking phase", 9: "cycles walk's end -> kernel's end", 10: "cycles kernel start -> candidates tested"}
ap = argparse.ArgumentParser()
ap.add_argument("--content", default="synthetic")
ap.add_argument("--k", type=int, default=0)
args = ap.parse_args()
if args.k == 0:
    for k in sorted(NAMES):
        lib = os.path.join(R, "lane_slam_amd", "liblanefront_ed%d.so" % k)
        if os.path.exists(lib):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--content", args.content, "--k", str(k)], env=dict(os.environ, LANEFRONT_LIBRARY=lib))
    sys.exit(0)
sys.path.insert(0, R)
import numpy as np, torch
from lane_slam_amd import FrontEnd, default_config, synth, _lib
B = 256
frames = synth.make_batch(B, seed0=0, threads=8)
if args.content == "real":
    real = np.load(os.path.join(R, "tests", "golden", "real_frames.npz"))
    rf = [real[k] for k in real.files if real[k].ndim == 3 and real[k].shape == frames.shape[1:]]
    frames = np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(B)])
dev = torch.device("cuda")
d = torch.from_numpy(frames).to(dev)
fe = FrontEnd(default_config("fullres"), max_frames=B, max_lines_per_color=256)
cap = B * 4096
out = {k: torch.zeros((cap, c) if c > 1 else cap, dtype={"f4": torch.float32, "i4": torch.int32, "u1": torch.uint8}[dt], device=dev) for k, dt, c in _lib.KEYLINE_FIELDS}
fo = torch.zeros(B + 1, dtype=torch.int32, device=dev)
s = _lib.LfKeylines(); s.capacity = cap; s.frame_offset = fo.data_ptr()
for k, _, _ in _lib.KEYLINE_FIELDS:
    setattr(s, k, out[k].data_ptr())
total = ctypes.c_int()
for _ in range(2):
    fe._check(fe.lib.lf_keylines_batch(fe.h, ctypes.c_void_p(d.data_ptr()), B, 0, 1, 1, None, ctypes.byref(s), 1, 1, ctypes.byref(total), None))
cnt = fe.keylines_fetch(0, 6, B)
v = cnt[:, 3].astype(np.int64) * (4 if args.k not in (4, 5, 6, 11) else 1)
print("%-40s mean %9.0f   max %9d (frame %d: %d anchors)   of the frame with most anchors (%d): %d" % (NAMES[args.k], v.mean(), v.max(), int(v.argmax()), cnt[int(v.argmax()), 0], cnt[:, 0].max(), v[int(cnt[:, 0].argmax())]))
