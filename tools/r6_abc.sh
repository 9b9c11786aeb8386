# headline for three library builds (A = liblanefront_A.so, B = product, C = liblanefront_C.so), alternating: bash tools/r6_abc.sh <rounds> <steps>
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${1:-3}); do for v in A B C; do
  case $v in A) export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so;; B) unset LANEFRONT_LIBRARY;; C) export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_C.so;; esac
  python3 $R/bench.py --gpus 1 --steps ${2:-20} --warmup 5 --secondary none --cpu-frames -1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done; done
