#!/usr/bin/env python3
"""Camera frames (or clutter) through D front-end handles in flight, nothing else: the run to put under
`rocprofv3 --kernel-trace` when the question is what the device does at the depth bench.py's content rows use.
    python3 tools/pipe_content.py [--depth 18] [--rounds 4] [--content real|clutter|synthetic]
tools/pipe_overlap.py reads the trace."""
import argparse, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lane_slam_amd import FrontEnd, default_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=18)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--content", default="real")
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, D, cap = args.batch, args.depth, 512
cfg = default_config("fullres")
fes = [FrontEnd(cfg, max_frames=B, max_lines_per_color=cap) for _ in range(D)]
if args.content == "real":
    z = np.load(os.path.join(ROOT, "tests", "golden", "real_jpegs.npz"))
    streams = [bytes(z["jpeg%02d" % k]) for k in range(len(z["names"]))]
    fr, st = fes[0].decode_jpeg_batch(streams, n_threads=4)
    rf = [fr[k] for k in range(len(streams)) if st[k] == 0]
    frames = np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(B)])
else:
    frames = synth.make_batch(B, 0)
    if args.content == "clutter":
        rng = np.random.default_rng(4321)
        for img in frames:
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
dev = torch.device("cuda:0")
d = torch.from_numpy(np.ascontiguousarray(frames)).to(dev)
outs = []
capn = B * 3 * cap // 8                 # segments of a batch the outputs hold (bench.py's alloc_out; ~100 per frame on these frames)
for _ in range(D):
    outs.append({"frame_offset": torch.zeros(B + 1, dtype=torch.int32, device=dev), "lines": torch.zeros(capn, 4, dtype=torch.float32, device=dev),
                 "normals": torch.zeros(capn, 2, dtype=torch.float32, device=dev), "color": torch.zeros(capn, dtype=torch.uint8, device=dev),
                 "pixels_normalized": torch.zeros(capn, 4, dtype=torch.float32, device=dev), "ground": torch.zeros(capn, 4, dtype=torch.float64, device=dev),
                 "keep": torch.zeros(capn, dtype=torch.uint8, device=dev), "desc": torch.zeros(capn, 72, dtype=torch.float32, device=dev),
                 "code": torch.zeros(capn, 32, dtype=torch.uint8, device=dev)})
ptrs = [{k: v.data_ptr() for k, v in o.items()} for o in outs]
torch.cuda.synchronize()


def go(nb):
    inflight, seg = [], 0
    for k in range(nb):
        slot = k % D
        if len(inflight) == D:
            seg += fes[inflight.pop(0)].wait()
        fes[slot].submit_device(d.data_ptr(), B, ptrs[slot], capn, describe=True)
        inflight.append(slot)
    while inflight:
        seg += fes[inflight.pop(0)].wait()
    return seg


go(2 * D)
torch.cuda.synchronize()
t0 = time.perf_counter()
nb = args.rounds * D
seg = go(nb)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s: %d in flight, %d batches of %d frames: %.1f frames/s, %.3f ms per batch, %.1f segments per frame" % (
    args.content, D, nb, B, nb * B / dt, dt / nb * 1e3, seg / (nb * B)))
