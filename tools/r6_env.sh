# driver-form headline under environment variants: bash tools/r6_env.sh <rounds>
R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --secondary none --cpu-frames -1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in $(seq 1 ${1:-3}); do
  unset LANEFRONT_LIBRARY LF_GROW_BITMAP; run base
  export LF_GROW_BITMAP=8192; run used8k
  export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so; run used8k_reg128
  export LF_GROW_BITMAP=4096; run used4k_reg128
done
