R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="--secondary none --cpu-frames -1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_i1 -- python3 $R/bench.py --steps 2 --warmup 1 $B --depth 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/pmc_a1 -- python3 $R/bench.py --steps 2 --warmup 1 $B --depth 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/pmc_a6 -- python3 $R/bench.py --steps 12 --warmup 6 $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_w6 -- python3 $R/bench.py --steps 12 --warmup 6 $B > /dev/null 2>&1
find $R/gpurun_out/pmc_* -name "*counter_collection.csv" | xargs ls -la
