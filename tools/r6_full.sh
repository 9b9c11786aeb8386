# full default bench (with the secondary rows) for A (liblanefront_A.so) and B (product): bash tools/r6_full.sh <tag> [A|B|AB]
R=$GRAFT_REPO_ROOT
T=${1:-r06full}; W=${2:-AB}; mkdir -p $R/gpurun_out/$T
for v in A B; do
  case $W in *$v*) ;; *) continue;; esac
  if [ $v = A ]; then export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so; else unset LANEFRONT_LIBRARY; fi
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/$T/bench_$v.json 2>$R/gpurun_out/$T/bench_$v.err
  python3 - $R/gpurun_out/$T/bench_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); s = d.get("secondary", {})
print(sys.argv[2], "value", d["value"], "ms", d["ms_per_step"], "parity", (d.get("parity_gate") or {}).get("identical"))
for k in ("real_frames", "clutter_frames", "jpeg_ingest", "stream_configs2", "edlines_keylines_pipelined", "assoc_stress_configs4"):
    v = s.get(k)
    if isinstance(v, dict): print("  ", k, v.get("value"), {kk: vv for kk, vv in v.items() if kk in ("one_feeder", "tie_pass_ms", "ms")})
for k in d.get("kernels", [])[:30]:
    print("   kernel", k)
PY
done
