# bench.py on the other BASELINE / model configurations: profiles/r02_other_configs.log.
mkdir -p gpurun_out
run() { timeout 300 python bench.py --no-cpu-baseline --no-parity --no-profile --steps 40 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['value'], d['sequential_value'])"; }
run --batch 1024 --refine-iters 2
run --natural-exit
run --precision fp32 --batch 128
run --model parseq-tiny
run --model vitstr
run --model parseq-patch16-224 --batch 64
run --batch 256
run --batch 128
