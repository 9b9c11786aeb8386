set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r1f
python -m pytest tests -m gpu -x -q > $R/gpurun_out/r1f/pytest.log 2>&1; echo "pytest rc=$?"
python bench.py --steps 30 --warmup 6 > $R/gpurun_out/r1f/bench_n1.json 2> $R/gpurun_out/r1f/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1f/d3 -- python3 $R/bench.py --steps 9 --warmup 3 --cpu-frames -1 > $R/gpurun_out/r1f/bench_d3_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1f/d1 -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-frames -1 --depth 1 > $R/gpurun_out/r1f/bench_d1_rocprof.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r1f/pw -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames -1 --depth 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r1f/pf -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames -1 --depth 1 > /dev/null 2>&1
find $R/gpurun_out/r1f -name "*.csv" | head -30
tail -3 $R/gpurun_out/r1f/pytest.log; cat $R/gpurun_out/r1f/bench_n1.json
