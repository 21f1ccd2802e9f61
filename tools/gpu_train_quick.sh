# Training GPU tests and tools/train_bench.py (bf16-operand mode), without the rocprof pass
timeout 900 python -m pytest tests/test_training.py -m gpu -q --timeout 600 2>&1 | tail -2
timeout 600 python tools/train_bench.py --steps 3 --warmup 1 2>/dev/null | cut -c1-230
