import time, numpy as np, sys
sys.path.insert(0,'.')
from lane_slam_amd import LineDetectorHIP, synth
from lane_slam_amd.config import DEFAULT_DETECTOR_CONFIGURATION
det = LineDetectorHIP(dict(DEFAULT_DETECTOR_CONFIGURATION))
imgs=[np.ascontiguousarray(synth.make_frame(i)[::4, ::4][40:]) for i in range(16)]
for im in imgs[:4]:
    det.setImage(im); [det.detectLines(c) for c in ("white","yellow","red")]
fe=det._fe
fe.reset_timing(); fe.set_profiling(True)
ts=[];td=[]
for i in range(100):
    t0=time.perf_counter(); det.setImage(imgs[i%16]); t1=time.perf_counter()
    for c in ("white","yellow","red"): det.detectLines(c)
    t2=time.perf_counter(); ts.append(t1-t0); td.append(t2-t1)
fe.set_profiling(False)
print("setImage median ms", 1e3*np.median(ts), "3x detectLines median ms", 1e3*np.median(td))
for k,(ms,n) in fe.timing().items():
    if n: print("  %-40s %.4f ms x %d" % (k, ms/n, n))
