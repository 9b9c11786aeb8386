# Device memory a handle holds, allocation by allocation (LF_ALLOC_TRACE=1): python tools/handle_footprint.py 2> alloc.txt
import os, sys
os.environ["LF_ALLOC_TRACE"]="1"
import torch
sys.path.insert(0, ".")
from lane_slam_amd import FrontEnd, default_config
for geo, nf, ins in (("vga", 256, None), ("hd", 16, (1080, 1920))):
    sys.stderr.write("===== %s %d\n" % (geo, nf)); sys.stderr.flush()
    free0 = torch.cuda.mem_get_info()[0]
    cfg = default_config("fullres", in_size=ins) if ins else default_config("fullres")
    fe = FrontEnd(cfg, max_frames=nf, device=0)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    sys.stderr.write("===== %s: handle holds %.2f GB (lsd %dx%d)\n" % (geo, (free0 - free1) / 1e9, fe.lsd_cols, fe.lsd_rows)); sys.stderr.flush()
    fe.close()
