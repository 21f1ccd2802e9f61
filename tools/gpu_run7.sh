set -x
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -m gpu -q --timeout 900 -k "encoder_blocks or fused" > gpurun_out/r2_ops.log 2>&1; tail -8 gpurun_out/r2_ops.log
python bench.py --no-cpu-baseline --steps 50 --exact-precision bf16x3 > gpurun_out/r2_bench_blocks.log 2> gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench_blocks.log
python -m pytest tests -m gpu -q --maxfail 12 --timeout 900 > gpurun_out/r2_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_gpu_tests.log
grep -v "^  File" gpurun_out/r2_gpu_tests.log | tail -40
