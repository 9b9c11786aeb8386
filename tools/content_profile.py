"""One handle, camera-frame content (tests/golden/real_jpegs.npz tiled to a 256-frame batch), a few batches back to back: run under
rocprofv3 --kernel-trace --stats for the per-kernel times on busy content (profiles/*_real_kernel_stats.csv).
python tools/content_profile.py [real|clutter] [batches]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_slam_amd import FrontEnd, default_config, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "real"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = 256
cfg = default_config("fullres")
if os.environ.get("SEED_ORDER"): cfg["lsd"]["seed_order"] = os.environ["SEED_ORDER"]
fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=512)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if kind == "real":
    zj = np.load(os.path.join(ROOT, "tests", "golden", "real_jpegs.npz"))
    streams = [bytes(zj["jpeg%02d" % k]) for k in range(len(zj["names"]))]
    rf, st_ = fe.decode_jpeg_batch(streams, n_threads=4)
    rf = [rf[k] for k in range(len(streams)) if st_[k] == 0 and rf[k].shape == (480, 640, 3)]
    batch = np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(B)])
else:
    batch = synth.make_batch(B, seed0=0)
    rng_c = np.random.default_rng(4321)
    for f_ in range(B):
        img = batch[f_]
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
d = torch.from_numpy(np.ascontiguousarray(batch)).cuda()
torch.cuda.synchronize()
n = 0
import bench
out = bench.alloc_out(torch, torch.device("cuda:0"), B, B * 3 * 512)
ptrs = {k: v.data_ptr() for k, v in out.items()}
for _ in range(nb):
    fe.submit_device(d.data_ptr(), B, ptrs, B * 3 * 512, describe=True)
    n = fe.wait()
print(kind, "segments per batch", n, "lists", fe.lsd_list_capacity())
