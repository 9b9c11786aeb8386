# One profiling round on the GPU box: bash tools/profile_round.sh <tag>   (results under gpurun_out/<tag>/)
set -x
R=$GRAFT_REPO_ROOT
T=${1:-r02a}; mkdir -p $R/gpurun_out/$T
python -m pytest tests -m gpu -x -q > $R/gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?"
python bench.py > $R/gpurun_out/$T/bench_n1.json 2> $R/gpurun_out/$T/bench_n1.err
# the driver's form of the same run (20 timed steps: the six-deep pipeline's fill and drain weigh more)
python bench.py --gpus 1 --steps 20 --warmup 5 --secondary none > $R/gpurun_out/$T/bench_driver_form.json 2>/dev/null
python tools/assoc_rate.py --tie-rule lowest > $R/gpurun_out/$T/assoc_rate.txt 2>&1
python tools/assoc_rate.py --tie-rule mihasher >> $R/gpurun_out/$T/assoc_rate.txt 2>&1
LF_TIE_SHAPE=big python tools/assoc_rate.py --tie-rule mihasher --pairs 4096x50000,16384x50000 >> $R/gpurun_out/$T/assoc_rate.txt 2>&1
LF_TIE_SHAPE=small python tools/assoc_rate.py --tie-rule mihasher --pairs 4096x50000,16384x50000 >> $R/gpurun_out/$T/assoc_rate.txt 2>&1
# round 5: the A/B options beside the defaults (opencv32 + mihasher), same call
for ab in "opencv32 lowest" "opencv30 mihasher" "opencv30 lowest"; do set -- $ab; python bench.py --steps 100 --secondary none --cpu-frames -1 --seed-order $1 --tie-rule $2 > $R/gpurun_out/$T/bench_ab_$1_$2.json 2>/dev/null; done
python bench.py --geometry hd --steps 30 --secondary none --cpu-frames -1 > $R/gpurun_out/$T/bench_hd.json 2>/dev/null
if [ -f lane_slam_amd/liblanefront_sstamps.so ]; then LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_sstamps.so python tools/seed_stamps.py > $R/gpurun_out/$T/seed_stamps.txt 2>&1; fi
# round 6: the growing kernel's phase stamps on the three kinds of content (diagnostic build), the content rows with growing left out
if [ -f lane_slam_amd/liblanefront_stamps.so ]; then
  for m in SYNTHETIC REAL CLUTTER; do env LF_STAMPS_$m=1 LANEFRONT_LIBRARY=$R/lane_slam_amd/liblanefront_stamps.so python tools/grow_stamps.py 64 > $R/gpurun_out/$T/grow_stamps_$m.txt 2>&1; done
fi
for c in real clutter; do for sk in "" grow; do echo -n "$c skip=[$sk] "; LF_DIAG_SKIP=$sk python tools/pipe_content.py --content $c --rounds 4 2>&1 | tail -1; done; done > $R/gpurun_out/$T/content_whatif.txt 2>&1
LF_ALLOC_TRACE=1 python tools/handle_footprint.py 2> $R/gpurun_out/$T/handle_footprint.txt
bash tools/whatif_round.sh > $R/gpurun_out/$T/whatif.txt 2>&1
python tools/assoc_rate.py --gating --pairs 16384x50000 >> $R/gpurun_out/$T/assoc_rate.txt 2>&1
LF_ASSOC_INT8=1 python tools/assoc_rate.py --pairs 16384x50000,65536x262144 >> $R/gpurun_out/$T/assoc_rate.txt 2>&1
LF_ASSOC_INT8=1 python tools/assoc_rate.py --gating --pairs 16384x50000 >> $R/gpurun_out/$T/assoc_rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="--secondary none --cpu-frames -1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/d6 -- python3 $R/bench.py --steps 12 --warmup 4 $B > $R/gpurun_out/$T/bench_d6_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/d1 -- python3 $R/bench.py --steps 6 --warmup 2 $B --depth 1 > $R/gpurun_out/$T/bench_d1_rocprof.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/pw -- python3 $R/bench.py --steps 2 --warmup 1 $B --depth 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pf -- python3 $R/bench.py --steps 2 --warmup 1 $B --depth 1 > /dev/null 2>&1
# wave-slot occupancy and vector issue utilisation of the LSD region growing kernel: one batch in flight
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/$T/pi -- python3 $R/bench.py --steps 2 --warmup 1 $B --depth 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$T/pq -- python3 $R/bench.py --steps 2 --warmup 1 $B --depth 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/$T/pq2 -- python3 $R/bench.py --steps 2 --warmup 1 $B --depth 1 > /dev/null 2>&1
# the same with six batches in flight (the configuration the headline is measured in)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$T/pq6 -- python3 $R/bench.py --steps 12 --warmup 4 $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/$T/pq26 -- python3 $R/bench.py --steps 12 --warmup 4 $B > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/$T/pa -- python3 $R/tools/assoc_rate.py --pairs 16384x50000 --reps 5 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/pg -- python3 $R/tools/assoc_rate.py --pairs 16384x50000 --reps 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/as -- python3 $R/tools/assoc_rate.py --pairs 4096x50000,16384x50000 --reps 5 > /dev/null 2>&1
# round 6: the second detector and the ingest stage: kernel times (rocprofv3), the walking wave's cycle budget (diagnostic builds), passes
# per frame of the Huffman decoder, the ingest rates
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/kl -- python3 $R/tools/keylines_rate.py --octaves 1,3 > $R/gpurun_out/$T/keylines_rate.txt 2>/dev/null
for c in real clutter; do python3 $R/tools/keylines_rate.py --octaves 1,3 --content $c 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/$T/keylines_rate.txt; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/jp -- python3 $R/tools/ingest_rate.py --threads 8 --entropy gpu --depth 8 --steps 64 --quality 80 --feeders 2 > $R/gpurun_out/$T/ingest_rate.txt 2>/dev/null
python3 $R/tools/ingest_rate.py --threads 8 --entropy gpu --depth 8 --steps 64 --quality 80 --feeders 1 2>/dev/null | grep "entropy gpu" | sed -e 's/$/   (one feeder)/' >> $R/gpurun_out/$T/ingest_rate.txt
LF_JH_SPLIT=1 python3 $R/tools/ingest_rate.py --threads 8 --entropy gpu --depth 8 --steps 64 --quality 80 --feeders 2 2>/dev/null | grep "entropy gpu" | sed -e 's/$/   (LF_JH_SPLIT=1: k_jh_unstuff in front)/' >> $R/gpurun_out/$T/ingest_rate.txt
cd $R
python3 tools/jpeg_host_time.py 2>/dev/null | grep "host time" >> $R/gpurun_out/$T/ingest_rate.txt
LF_JH_DEBUG=3 python3 tools/jh_passes.py 2>&1 | grep "passes\|pass 0\|unit position" > $R/gpurun_out/$T/jh_passes.txt
if [ -f lane_slam_amd/liblanefront_ed1.so ]; then python3 tools/ed_stamps.py 2>/dev/null | grep -v amdgpu.ids > $R/gpurun_out/$T/ed_stamps.txt; python3 tools/ed_stamps.py --content real 2>/dev/null | grep -v amdgpu.ids >> $R/gpurun_out/$T/ed_stamps.txt; fi
find $R/gpurun_out/$T -name "*.csv" | head -40
tail -3 $R/gpurun_out/$T/pytest.log; cat $R/gpurun_out/$T/assoc_rate.txt; tail -c 3000 $R/gpurun_out/$T/bench_n1.json
