R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --gpus 1 --steps ${STEPS:-100} --warmup 8 --secondary none --cpu-frames -1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
export LF_ASSOC_SHAPE=small
for rep in $(seq 1 ${1:-2}); do
  for sl in 512 768 1024 1536 2048; do export LF_ASSOC_SLOTS=$sl; run slots$sl; done
done
