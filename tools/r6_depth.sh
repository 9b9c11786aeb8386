R=$GRAFT_REPO_ROOT
for rep in 1 2; do for d in 6 7 8 9 10; do
  python3 $R/bench.py --gpus 1 --steps 160 --warmup 16 --secondary none --cpu-frames -1 --depth $d 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth $d', d['value'], d['ms_per_step'])"
done; done
