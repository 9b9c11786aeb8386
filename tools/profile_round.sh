set -x
R=$GRAFT_REPO_ROOT
T=${1:-r1f}; mkdir -p $R/gpurun_out/$T
python -m pytest tests -m gpu -x -q > $R/gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?"
python bench.py --steps 30 --warmup 6 > $R/gpurun_out/$T/bench_n1.json 2> $R/gpurun_out/$T/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/d3 -- python3 $R/bench.py --steps 9 --warmup 3 --cpu-frames -1 > $R/gpurun_out/$T/bench_d3_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/d1 -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-frames -1 --depth 1 > $R/gpurun_out/$T/bench_d1_rocprof.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/pw -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames -1 --depth 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pf -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames -1 --depth 1 > /dev/null 2>&1
find $R/gpurun_out/$T -name "*.csv" | head -30
tail -3 $R/gpurun_out/$T/pytest.log; cat $R/gpurun_out/$T/bench_n1.json
