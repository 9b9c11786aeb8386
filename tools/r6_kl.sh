# same-box A/B of the pipelined KeyLines row (A = lane_slam_amd/liblanefront_A.so): bash tools/r6_kl.sh
R=$GRAFT_REPO_ROOT
cat > /tmp/klp.py <<'PY'
import os, sys, time, ctypes as ct
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from lane_slam_amd import FrontEnd, default_config, synth, _lib as L
B, D = 256, 8
dev = torch.device("cuda")
host = synth.make_batch(B, seed0=0, threads=8)
d = torch.from_numpy(host).to(dev)
fes = [FrontEnd(default_config("fullres"), max_frames=B, max_lines_per_color=256) for _ in range(D)]
kcap = B * 512
kouts = [{k: torch.zeros((kcap, c) if c > 1 else kcap, dtype={"f4": torch.float32, "i4": torch.int32, "u1": torch.uint8}[dt], device=dev) for k, dt, c in L.KEYLINE_FIELDS} for _ in range(D)]
kfos = [torch.zeros(B + 1, dtype=torch.int32, device=dev) for _ in range(D)]
kptrs = [dict({k: v.data_ptr() for k, v in o.items()}, frame_offset=f.data_ptr()) for o, f in zip(kouts, kfos)]
def run(n):
    infl = []
    for k in range(n):
        s = k % D
        if len(infl) == D: fes[infl.pop(0)].wait()
        fes[s].keylines_submit_device(d.data_ptr(), B, kptrs[s], kcap, n_octaves=3)
        infl.append(s)
    while infl: fes[infl.pop(0)].wait()
run(D); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); run(8 * D); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s keylines pipelined: %.1f k frames/s" % (os.environ.get("TAG", ""), 8 * D * B / dt / 1e3))
PY
for rep in 1 2; do for v in A B; do
  if [ $v = A ]; then export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so; else unset LANEFRONT_LIBRARY; fi
  TAG=$v python3 /tmp/klp.py 2>/dev/null | tail -3
done; done
