R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --gpus 1 --steps 100 --warmup 8 --secondary none --cpu-frames -1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in $(seq 1 ${1:-3}); do
  unset LF_GROW_PAD_KB; run base
  export LF_GROW_PAD_KB=32; run pad32_4perCU
  export LF_GROW_PAD_KB=40; run pad40_3perCU
done
