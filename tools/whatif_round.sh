# What each stage costs the PIPELINED step: the default bench with one stage skipped (LF_DIAG_SKIP: results are wrong, timing only) and the A/B configurations
for v in "" grow seedchain seeddense; do
  export LF_DIAG_SKIP=$v
  echo -n "skip '$v': "; python bench.py --steps 60 --secondary none --cpu-frames -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
unset LF_DIAG_SKIP
echo -n "opencv30+mihasher: "; python bench.py --steps 60 --secondary none --cpu-frames -1 --seed-order opencv30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "opencv32+lowest: "; python bench.py --steps 60 --secondary none --cpu-frames -1 --tie-rule lowest 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
