#!/usr/bin/env python3
"""How long does FrontEnd.wait() take when the batch is long done?  And submit_device()?  (host-side cost of a step)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32"); os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lane_slam_amd import FrontEnd, default_config, synth
B, cap = 256, 512
cfg = default_config("fullres")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fes = [FrontEnd(cfg, max_frames=B, max_lines_per_color=cap) for _ in range(D)]
dev = torch.device("cuda:0")
frames = torch.from_numpy(synth.make_batch(B, 0, threads=16)).to(dev)
capn = B * 3 * cap // 8
def alloc():
    return {"frame_offset": torch.zeros(B + 1, dtype=torch.int32, device=dev), "lines": torch.zeros(capn, 4, dtype=torch.float32, device=dev),
            "normals": torch.zeros(capn, 2, dtype=torch.float32, device=dev), "color": torch.zeros(capn, dtype=torch.uint8, device=dev),
            "pixels_normalized": torch.zeros(capn, 4, dtype=torch.float32, device=dev), "ground": torch.zeros(capn, 4, dtype=torch.float64, device=dev),
            "keep": torch.zeros(capn, dtype=torch.uint8, device=dev), "desc": torch.zeros(capn, 72, dtype=torch.float32, device=dev),
            "code": torch.zeros(capn, 32, dtype=torch.uint8, device=dev)}
outs = [alloc() for _ in range(D)]
ptrs = [{k: v.data_ptr() for k, v in o.items()} for o in outs]
for i in range(D):
    fes[i].submit_device(frames.data_ptr(), B, ptrs[i], capn, describe=True); fes[i].wait()
torch.cuda.synchronize()
ts, tw = [], []
for rep in range(5):
    for i in range(D):
        t0 = time.perf_counter(); fes[i].submit_device(frames.data_ptr(), B, ptrs[i], capn, describe=True); ts.append(time.perf_counter() - t0)
    time.sleep(0.2)                      # everything is long done
    for i in range(D):
        t0 = time.perf_counter(); fes[i].wait(); tw.append(time.perf_counter() - t0)
print("submit_device: median %.3f ms  max %.3f | wait() on a finished batch: median %.3f ms max %.3f" % (1e3 * np.median(ts), 1e3 * max(ts), 1e3 * np.median(tw), 1e3 * max(tw)))
# throughput with nothing but submit / wait (no association)
def go(nb):
    infl = []
    for k in range(nb):
        s = k % D
        if len(infl) == D: fes[infl.pop(0)].wait()
        fes[s].submit_device(frames.data_ptr(), B, ptrs[s], capn, describe=True); infl.append(s)
    while infl: fes[infl.pop(0)].wait()
go(2 * D); torch.cuda.synchronize()
t0 = time.perf_counter(); go(12 * D); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("front end alone, %d in flight: %.1f frames/s, %.3f ms per batch" % (D, 12 * D * B / dt, 1e3 * dt / (12 * D)))
