import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from lane_slam_amd import FrontEnd, default_config, synth
cfg = default_config("fullres"); cfg["lsd"]["seed_order"] = "opencv32"
fe = FrontEnd(cfg, max_frames=2, max_lines_per_color=512)
frames = synth.make_batch(2, 0)
fe.process_batch(frames)
fe.process_batch(frames)
fe.close()
