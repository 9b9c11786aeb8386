# pipelined bench of library variants against the product build, alternating: bash tools/r6_variants.sh <name> [<name> ...]   (lane_slam_amd/liblanefront_<name>.so)
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in product "$@"; do
  if [ $v = product ]; then unset LANEFRONT_LIBRARY; else export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_$v.so; fi
  python3 $R/bench.py --steps 100 --secondary none --cpu-frames -1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done; done
