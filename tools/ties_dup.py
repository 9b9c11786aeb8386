"""The tie pass on maps that hold copies of the queries (exact: the bench steady state; one bit away: 28 equally near copies each) and on
random codes: lowest-index rule against the Mihasher rule, per-call time of lf_map_associate (HIP events on the map stream)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from lane_slam_amd import LineAssociator, synth
torch.cuda.init()
n, copies = 2370, 28
base = synth.random_codes(n, 1)
for kind in ("dup", "near", "random"):
    if kind == "dup":
        mapc = np.tile(base, (copies, 1))
    elif kind == "near":
        rng = np.random.default_rng(1)
        mapc = np.tile(base, (copies, 1)).copy()
        flip = rng.integers(0, 256, mapc.shape[0])
        mapc[np.arange(mapc.shape[0]), flip >> 3] ^= (1 << (flip & 7)).astype(np.uint8)      # every copy one bit away: ties at distance 1
    else:
        mapc = synth.random_codes(n * copies, 2)
    m = mapc.shape[0]
    for rule in ("lowest", "mihasher"):
        am = LineAssociator(capacity=m, color_gating=False, kept_only=False, tie_rule=rule)
        am.seed(mapc, np.zeros(m, np.uint8))
        q = torch.from_numpy(base).cuda(); qc = torch.zeros(n, dtype=torch.uint8, device="cuda")
        idx = torch.zeros(n, dtype=torch.int32, device="cuda"); dist = torch.zeros(n, dtype=torch.float32, device="cuda")
        for _ in range(3): am.associate_device(None, q.data_ptr(), qc.data_ptr(), n, idx.data_ptr(), dist.data_ptr())
        am.synchronize(); am.timing(); am.set_profiling(True)
        for _ in range(20): am.associate_device(None, q.data_ptr(), qc.data_ptr(), n, idx.data_ptr(), dist.data_ptr())
        am.synchronize(); t = am.timing()
        print(kind, rule, {k: round(v[0] / max(v[1], 1), 4) for k, v in t.items() if v[1] > 0})
        am.close()
