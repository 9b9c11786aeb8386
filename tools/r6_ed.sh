# same-box A/B of the EDLines path: bash tools/r6_ed.sh [soak frames]   (A = lane_slam_amd/liblanefront_A.so, B = the product build)
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_edlines.py tests/test_gpu_descriptor_params.py tests/test_gpu_lsd_keylines.py -m gpu -x -q 2>&1 | tail -2
if [ "${1:-0}" != "0" ]; then timeout 900 python $R/tools/soak_edlines.py --frames $1 2>&1 | tail -3; fi
for rep in 1 2; do for v in A B; do
  if [ $v = A ]; then export LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_A.so; else unset LANEFRONT_LIBRARY; fi
  for c in synthetic real clutter; do echo -n "$v $c: "; python3 $R/tools/keylines_rate.py --octaves 1,3 --content $c 2>/dev/null | grep "^octaves" | sed -e 's/KeyLines per.*synchronous call);//' | tr '\n' '|'; echo; done
done; done
