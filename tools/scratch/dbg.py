import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle.oracle import Oracle
from lane_slam_amd import default_config, synth
from lane_slam_amd.line_detector_hip import LineDetectorHIP
sys.path.insert(0, "tests")
from test_gpu_parity import DEFAULT_DETECTOR_CONFIGURATION
cfg = default_config("parity"); o = Oracle(cfg)
det = LineDetectorHIP(dict(DEFAULT_DETECTOR_CONFIGURATION))
for seed in (0, 4, 7):
    work = o.preprocess(synth.make_frame(seed)); det.setImage(work)
    bw = o.color_masks(o.bgr2hsv(work)); edges = o.canny(work)
    for ci, color in enumerate(("white", "yellow", "red")):
        d = det.detectLines(color); area = o.dilate(bw[ci]); lines = o.lsd(area & edges)
        if len(lines) == 0: continue
        ol, on, oc = o.find_normals(area, lines)
        dl = np.asarray(d.lines)
        print(seed, color, dl.shape, ol.shape)
        if dl.shape == ol.shape:
            bad = np.where((dl != ol).any(1))[0]
            for b in bad: print("  row", b, dl[b], ol[b], dl[b].view(np.uint32) - ol[b].view(np.uint32))
        else:
            print(dl); print(ol)
