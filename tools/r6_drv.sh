# driver-form headline, alternating A / B, n rounds: bash tools/r6_drv.sh <rounds>
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${1:-3}); do for v in A B; do
  if [ $v = A ]; then export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so; else unset LANEFRONT_LIBRARY; fi
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --secondary none --cpu-frames -1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done; done
