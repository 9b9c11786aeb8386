# the driver's 20-step form at several pipeline depths: bash tools/r6_depth20.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for d in 4 5 6 7 8; do
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --secondary none --cpu-frames -1 --depth $d 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth $d', d['value'], d['ms_per_step'])"
done; done
