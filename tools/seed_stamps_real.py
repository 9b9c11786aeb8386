"""The same as tools/seed_stamps.py on two of the real camera frames (tests/golden/real_jpegs.npz)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from lane_slam_amd import FrontEnd, default_config, synth
cfg = default_config("fullres"); cfg["lsd"]["seed_order"] = "opencv32"
fe = FrontEnd(cfg, max_frames=2, max_lines_per_color=512)
zj = np.load("tests/golden/real_jpegs.npz")
streams = [bytes(zj["jpeg%02d" % k]) for k in (3, 11)]
rf, st_ = fe.decode_jpeg_batch(streams, n_threads=2)
frames = np.stack(rf)
fe.process_batch(frames)
fe.process_batch(frames)
fe.close()
