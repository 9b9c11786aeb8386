#!/usr/bin/env python3
"""Host time of one lf_jpeg_decode_batch_gpu call (header parsing, staging copy, H2D and launches; the GPU works behind it) and of one
submit_device, per 256-frame batch: what ONE feeder thread can sustain.  python tools/jpeg_host_time.py [--threads 8]"""
import argparse, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from lane_slam_amd import FrontEnd, default_config, synth
ap = argparse.ArgumentParser(); ap.add_argument("--threads", type=int, default=8); args = ap.parse_args()
B = 256
streams = []
for i in range(32):
    b = io.BytesIO(); Image.fromarray(synth.make_frame(i)[..., ::-1].copy()).save(b, "JPEG", quality=80, subsampling=2); streams.append(b.getvalue())
msgs = [streams[i % 32] for i in range(B)]
fes = [FrontEnd(default_config("fullres"), max_frames=B, max_lines_per_color=512) for _ in range(4)]
bufs = [fe.frames_buffer()[0] for fe in fes]
for fe, b in zip(fes, bufs):
    fe.decode_jpeg_batch(msgs, n_threads=args.threads, device_ptr=b, entropy="gpu")
torch.cuda.synchronize()
for rep in range(3):
    ts = []
    for k in range(4):
        t0 = time.perf_counter()
        fes[k].decode_jpeg_batch(msgs, n_threads=args.threads, device_ptr=bufs[k], entropy="gpu")
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("decode_jpeg_batch host time per call (4 handles, GPU idle at the start): %s ms" % " ".join("%.2f" % (1e3 * t) for t in ts))
