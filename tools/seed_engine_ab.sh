# The dense phase of k_lsd_seed32 in its level-synchronous form (the block engine, -DLF_SEED_ENGINE=1) against the default wave form:
# builds lane_slam_amd/liblanefront_engine.so, runs the seed-order tests on it and both benches.  On the GPU box:
#   bash tools/seed_engine_ab.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
make -C $R/lane_slam_amd/csrc EXTRA=-DLF_SEED_ENGINE=1 BUILD=_build_engine OUT=../liblanefront_engine.so -j8 > /dev/null
LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_engine.so python -m pytest $R/tests/test_gpu_seed_order.py $R/tests/test_gpu_footprint.py -q -m gpu 2>&1 | grep -E "passed|failed"
for lib in "" $R/lane_slam_amd/liblanefront_engine.so; do
  if [ -n "$lib" ]; then export LANEFRONT_LIBRARY=$lib; else unset LANEFRONT_LIBRARY; fi
  echo -n "library '${lib:-default}': "; python $R/bench.py --steps 100 --secondary none --cpu-frames -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms per step')"
done
