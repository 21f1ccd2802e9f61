# One-rank self-test of the distributed path of bench.py (RCCL group + the real all-gather) beside the plain run: checks that two batches in flight still overlap with a communicator alive (GPU_MAX_HW_QUEUES).
for i in 1 2; do
timeout 300 python bench.py --force-dist --no-cpu-baseline --no-parity --no-profile --steps 40 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist (real 1-rank all-gather)', d['value'], d['sequential_value'])"
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-profile --steps 40 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['sequential_value'])"
done
