for q in "" 8 16; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python bench.py --force-dist --no-cpu-baseline --no-parity --no-profile --steps 40 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist GPU_MAX_HW_QUEUES=$q', d['value'], d['sequential_value'])"
  timeout 300 python bench.py --no-cpu-baseline --no-parity --no-profile --steps 40 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain      GPU_MAX_HW_QUEUES=$q', d['value'], d['sequential_value'])"
done
