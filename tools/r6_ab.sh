# same-box A/B of two library builds: bash tools/r6_ab.sh <tag> [pytest -k expr|none|all]   (A = lane_slam_amd/liblanefront_A.so, B = the product build)
R=$GRAFT_REPO_ROOT
T=${1:-r06ab}; K=${2:-none}; KN=${3:-k_lsd_grow_bm}; mkdir -p $R/gpurun_out/$T
if [ "$K" != "none" ]; then
  if [ "$K" = "all" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/$T/pytest.log 2>&1; else timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > $R/gpurun_out/$T/pytest.log 2>&1; fi
  echo "pytest rc=$?"; tail -3 $R/gpurun_out/$T/pytest.log
fi
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so; else unset LANEFRONT_LIBRARY; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/d1_$v$rep -- python3 $R/bench.py --steps 6 --warmup 2 --secondary none --cpu-frames -1 --depth 1 > /dev/null 2>&1
  f=$(find $R/gpurun_out/$T/d1_$v$rep -name "*kernel_stats.csv" | head -1)
  echo -n "$v$rep $KN us: "; grep "$KN(" $f | awk -F'",' '{print $2}' | awk -F, '{print $3/1000}'
  python3 $R/bench.py --steps 60 --secondary none --cpu-frames -1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench', d['value'], d['ms_per_step'])"
done; done
