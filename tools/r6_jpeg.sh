# the device Huffman decoder: tests, passes per frame, kernel times and the ingest rate of the product build, then of
# lane_slam_amd/liblanefront_A.so: bash tools/r6_jpeg.sh [soak n]
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_real_frames.py -m gpu -x -q 2>&1 | tail -2
if [ "${1:-0}" != "0" ]; then timeout 900 python $R/tools/soak_jpeg.py --n $1 2>&1 | tail -2; fi
python3 $R/tools/jh_passes.py 2>&1 | grep "passes\|pass 0" | head -20
for v in B A; do
  if [ $v = A ]; then export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so; else unset LANEFRONT_LIBRARY; fi
  echo "== $v"; bash $R/tools/_jprof.sh 2>&1 | tail -8
done
