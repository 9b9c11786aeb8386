import os, sys
os.environ["LF_LSD_RECORDS"] = sys.argv[1] if len(sys.argv) > 1 else "1024"
sys.path.insert(0, ".")
from lane_slam_amd import FrontEnd, default_config, synth
cfg = default_config("fullres")
fe = FrontEnd(cfg, max_frames=2)
print(fe.lsd_list_capacity())
frames = synth.make_batch(2, seed0=77)
try:
    seg = fe.process_batch(frames, describe=True)
    print("ok", int(seg.frame_offset[-1]), fe.lsd_list_capacity())
except Exception as e:
    print("ERR", e)
