# quick A/B on the GPU box: bash tools/r6_quick.sh <tag> [pytest -k expression | none] [bench steps]
R=$GRAFT_REPO_ROOT
T=${1:-r06q}; K=${2:-none}; STEPS=${3:-60}; mkdir -p $R/gpurun_out/$T
if [ "$K" != "none" ]; then
  if [ "$K" = "all" ]; then timeout 600 python -m pytest tests -m gpu -x -q > $R/gpurun_out/$T/pytest.log 2>&1; else timeout 600 python -m pytest tests -m gpu -x -q -k "$K" > $R/gpurun_out/$T/pytest.log 2>&1; fi
  echo "pytest rc=$?"; tail -3 $R/gpurun_out/$T/pytest.log
fi
python bench.py --steps $STEPS --secondary none --cpu-frames -1 > $R/gpurun_out/$T/bench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/d1 -- python3 $R/bench.py --steps 6 --warmup 2 --secondary none --cpu-frames -1 --depth 1 > $R/gpurun_out/$T/bench_d1_rocprof.json 2>/dev/null
cd $R
f=$(find gpurun_out/$T/d1 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print("%-28s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
python -c "
import json
d=json.load(open('gpurun_out/$T/bench.json')); print('bench', d['value'], d['ms_per_step'], d.get('parity_gate'))"
