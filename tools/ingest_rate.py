#!/usr/bin/env python3
"""Ingest throughput: 640x480 JPEG streams -> BGR frames in HBM (lf_jpeg_decode_batch), alone
and followed by the whole front end, for a sweep of host thread counts.  Reported in DESIGN.md; never
bench.py's `value` (that is measured with frames already resident in HBM).

    python tools/ingest_rate.py [--threads 8,32,64] [--batch 256] [--steps 8]
"""
import argparse, io, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from lane_slam_amd import FrontEnd, default_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--threads", default="8,16,32,64,128")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--quality", type=int, default=75)
ap.add_argument("--entropy", default="gpu,host", help="entropy decoder(s): gpu (k_jhuff.hip) and / or host (threads)")
ap.add_argument("--feeders", type=int, default=1, help="host threads that each drive their own handles (ctypes releases the GIL inside the library)")
args = ap.parse_args()
B, D = args.batch, args.depth
torch.cuda.init()
dev = torch.device("cuda", 0)
uniq = 32
streams = []
for i in range(uniq):
    b = io.BytesIO()
    Image.fromarray(synth.make_frame(i)[..., ::-1].copy()).save(b, "JPEG", quality=args.quality, subsampling=2)
    streams.append(b.getvalue())
streams = (streams * ((B + uniq - 1) // uniq))[:B]
jpeg_bytes = sum(len(s) for s in streams) / B
cfg = default_config("fullres")
fes = [FrontEnd(cfg, max_frames=B, max_lines_per_color=512) for _ in range(D)]
cap = B * 3 * 512
outs = [{"frame_offset": torch.zeros(B + 1, dtype=torch.int32, device=dev), "lines": torch.zeros(cap, 4, dtype=torch.float32, device=dev),
         "keep": torch.zeros(cap, dtype=torch.uint8, device=dev), "ground": torch.zeros(cap, 4, dtype=torch.float64, device=dev),
         "code": torch.zeros(cap, 32, dtype=torch.uint8, device=dev)} for _ in range(D)]
ptrs = [{k: v.data_ptr() for k, v in o.items()} for o in outs]
bufs = [fe.frames_buffer()[0] for fe in fes]
print("640x480 4:2:0 q%d, %.1f KB per stream, batch %d, host cpus %d" % (args.quality, jpeg_bytes / 1024, B, os.cpu_count()))
for ent, nt in [(e, int(t)) for e in args.entropy.split(",") for t in args.threads.split(",")]:
    # decode only
    fes[0].decode_jpeg_batch(streams, n_threads=nt, device_ptr=bufs[0], entropy=ent); fes[0].synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fes[0].decode_jpeg_batch(streams, n_threads=nt, device_ptr=bufs[0], entropy=ent)
    fes[0].synchronize()
    dt = time.perf_counter() - t0
    dec = B * args.steps / dt
    # decode + front end, D handles in flight, split over --feeders host threads
    import threading
    F = max(1, min(args.feeders, D))
    def run(n, fi=0):
        mine = [sl for sl in range(D) if sl % F == fi]
        infl = []
        for k in range(n):
            sl = mine[k % len(mine)]
            if len(infl) == len(mine): fes[infl.pop(0)].wait()
            if ent == "gpu":                                   # queued (round 6): the feeder does not wait for the decoder
                fes[sl].decode_jpeg_batch_async(streams, device_ptr=bufs[sl], n_threads=max(1, nt // F), for_detect=not os.environ.get("LF_INGEST_WHOLE_FRAMES"))
            else:
                fes[sl].decode_jpeg_batch(streams, n_threads=max(1, nt // F), device_ptr=bufs[sl], entropy=ent)
            fes[sl].submit_device(bufs[sl], B, ptrs[sl], cap, describe=True)
            infl.append(sl)
        while infl: fes[infl.pop(0)].wait()
    def run_all(n):
        th = [threading.Thread(target=run, args=(n // F, fi)) for fi in range(F)]
        for x in th: x.start()
        for x in th: x.join()
    run_all(D)
    t0 = time.perf_counter(); run_all(args.steps); dt = time.perf_counter() - t0
    print("entropy %-4s threads %3d: decode %8.0f frames/s   decode + front end %8.0f frames/s" % (ent, nt, dec, B * args.steps / dt))
# device time of the ingest kernels (HIP events on the handle's stream), one batch in flight
fes[0].reset_timing(); fes[0].set_profiling(True)
for _ in range(5):
    fes[0].decode_jpeg_batch(streams, n_threads=8, device_ptr=bufs[0]); fes[0].synchronize()
fes[0].set_profiling(False)
ms, n = fes[0].timing()["jpeg(idct+upsample+color)"]
P = 480 * 640
alg = B * (P * 1.5 + P * 1.5 + P * 3)      # planes written + planes read + BGR written (coefficient lists come on top)
print("k_jpeg_idct + k_jpeg_color: %.3f ms per %d-frame batch -> %.0f GB/s of plane/pixel traffic" % (ms / n, B, alg / (ms / n) / 1e6))
