run() { LANEFRONT_LIBRARY=$GRAFT_REPO_ROOT/lane_slam_amd/$1 timeout 200 python bench.py --secondary none --cpu-frames -1 $3 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['stage']:x['avg_ms'] for x in d['kernels']}
print('$2', d['value'], d['ms_per_step'], 'canny', k['canny_nms'], 'pre', k['pre(resize+correct+hsv+masks+dilate)'], 'hyst', k['canny_hysteresis'])"; }
for i in 1 2 3; do run liblanefront.so NEW; run liblanefront_base.so BASE; done
