#!/usr/bin/env python3
"""End-to-end replay on one MI355X, the shape of BASELINE config 3: a stream of CompressedImage-like JPEG
messages -> batched ingest -> detect / describe / project / sanity -> association against a live map that
receives the kept segments (append-only, src/show_map/src/show_map.py:28-42) -> the three SegmentList topics'
wire bodies.  The first batch is checked against the oracle (pixels, segments, matches); the rest is timed.

    python tools/replay_demo.py [--frames 1024] [--batch 128] [--threads 32] [--geometry fullres|parity]
"""
import argparse, io, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from lane_slam_amd import FrontEnd, default_config, synth
from lane_slam_amd import segment_msgs as sm
from lane_slam_amd import LineAssociator

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--geometry", default="fullres", choices=["fullres", "parity"])
ap.add_argument("--map", type=int, default=50000)
args = ap.parse_args()
torch.cuda.init()
B = args.batch
cfg = default_config(args.geometry)

# "rosbag": JPEG streams of seeded synthetic camera frames (32 distinct, cycled)
streams = []
for i in range(32):
    b = io.BytesIO()
    Image.fromarray(synth.make_frame(9000 + i)[..., ::-1].copy()).save(b, "JPEG", quality=80, subsampling=2)
    streams.append(b.getvalue())
msgs = [streams[i % 32] for i in range(args.frames)]

fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=512)
dev_frames, _ = fe.frames_buffer()
live = LineAssociator(capacity=args.map + 65536, policy="append", kept_only=True)     # show_map.py:28-42: append the kept segments
live.seed(synth.random_codes(args.map, 1234))

# ---- first batch: check every stage against the oracle
from oracle.oracle import Oracle, jpeg_decode
o = Oracle(cfg)
st = fe.decode_jpeg_batch(msgs[:B], n_threads=args.threads, device_ptr=dev_frames)
assert not st.any()
seg = fe.process_batch(dev_frames, n_frames=B)
for f in (0, B // 2, B - 1):
    r = o.process_frame(jpeg_decode(msgs[f]))
    s = seg.frame(f)
    assert s.n == r["n"] and np.array_equal(s.lines, r["lines"]) and np.array_equal(s.keep, r["keep"]) and np.array_equal(s.code, r["code"])
idx, dist = live.associate(seg.code, seg.color)
oi, od = o.match(seg.code[:200], live.fetch(0, live.state()["size"])["code"])
assert np.array_equal(idx[:200], oi) and np.array_equal(dist[:200], od)
bodies, off = sm.serialize_segments(fe, seg, sm.FILTERED)
assert sm.split_segment_list(sm.segment_list_message(sm.header_bytes(0, 0, 0, "cam"), bodies[off[0]:off[1]]))[5].shape[0] == int(seg.keep[seg.frame_offset[0]:seg.frame_offset[1]].sum())
print("first batch verified against the oracle: %d segments in %d frames, %d kept" % (seg.n, B, int(seg.keep.sum())))

# ---- replay
t0 = time.perf_counter()
n_seg = n_kept = n_matched = wire = 0
for b0 in range(0, args.frames - B + 1, B):
    fe.decode_jpeg_batch(msgs[b0:b0 + B], n_threads=args.threads, device_ptr=dev_frames)
    seg = fe.process_batch(dev_frames, n_frames=B)
    if seg.n:
        d = {k: torch.from_numpy(np.ascontiguousarray(getattr(seg, k))).cuda() for k in ("frame_offset", "code", "color", "keep", "ground")}
        di = torch.zeros(seg.n, dtype=torch.int32, device="cuda")
        dd = torch.zeros(seg.n, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        live.step_device(None, {k: v.data_ptr() for k, v in d.items()}, seg.n, B, di.data_ptr(), dd.data_ptr(), step=b0 // B)
        live.synchronize()
        n_matched += int((di >= 0).sum().item())
        n_kept += int(seg.keep.sum())
    for stage in (sm.DETECTOR, sm.GROUND, sm.FILTERED):
        bodies, off = sm.serialize_segments(fe, seg, stage)
        wire += bodies.size
    n_seg += seg.n
dt = time.perf_counter() - t0
nb = (args.frames // B) * B
print("replayed %d frames in %.2f s: %.0f frames/s (synchronous, host-resident results, one handle); %d segments, %d kept, "
      "%d matched within 128 bits, map %d codes, %.1f MB of SegmentList bodies" % (nb, dt, nb / dt, n_seg, n_kept, n_matched, live.state()["size"], wire / 1e6))
