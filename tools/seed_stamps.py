"""Per-phase times of the seed-order kernels (k_lsd_seed32 / k_lsd_seed32_dense) on two synthetic lane frames; needs the diagnostic build:
    make -C lane_slam_amd/csrc EXTRA=-DLF_SEED_STAMPS BUILD=_build_sstamps OUT=../liblanefront_sstamps.so
    LANEFRONT_LIBRARY=$PWD/lane_slam_amd/liblanefront_sstamps.so python tools/seed_stamps.py"""
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
