# fused unstuffing A/B (LF_JH_SPLIT=1: the separate kernel): tests, soak, ingest rates
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_real_frames.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python $R/tools/soak_jpeg.py --n ${1:-300} 2>&1 | tail -1
for rep in 1 2; do for v in fused split; do
  if [ $v = split ]; then export LF_JH_SPLIT=1; else unset LF_JH_SPLIT; fi
  echo -n "$v: "; python3 $R/tools/ingest_rate.py --threads 8 --entropy gpu --depth 8 --steps 64 --quality 80 --feeders 2 2>&1 | grep "entropy gpu"
done; done
unset LF_JH_SPLIT
bash $R/tools/_jprof.sh 2>&1 | grep "^lf::"
