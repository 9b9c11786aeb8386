# k_lsd_grow phase stamps (diagnostic build) + the solo kernel times of a depth-1 bench: bash tools/r6_stamps.sh <tag>
R=$GRAFT_REPO_ROOT
T=${1:-r06x}; mkdir -p $R/gpurun_out/$T
export LF_LSD_RECORDS=full
LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_stamps.so python tools/grow_stamps.py 64 > $R/gpurun_out/$T/grow_stamps_synthetic.txt 2>&1
LF_STAMPS_REAL=1 LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_stamps.so python tools/grow_stamps.py 64 > $R/gpurun_out/$T/grow_stamps_real.txt 2>&1
unset LF_LSD_RECORDS
python bench.py --gpus 1 --steps 20 --warmup 5 --secondary none --cpu-frames -1 > $R/gpurun_out/$T/bench_driver_form.json 2>/dev/null
python bench.py --steps 100 --secondary none --cpu-frames -1 > $R/gpurun_out/$T/bench_100.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/d1 -- python3 $R/bench.py --steps 6 --warmup 2 --secondary none --cpu-frames -1 --depth 1 > $R/gpurun_out/$T/bench_d1_rocprof.json 2>/dev/null
cd $R
f=$(find gpurun_out/$T/d1 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]: print("%-28s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat gpurun_out/$T/grow_stamps_synthetic.txt; python -c "
import json,sys
for f in ('bench_driver_form','bench_100'):
    d=json.load(open('gpurun_out/$T/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
