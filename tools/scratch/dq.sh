for q in 8 12; do for d in 6 8 10; do
echo "queues=$q depth=$d"; GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 10 --cpu-frames -1 --depth $d | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done; done
