set -x
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -m gpu -q --timeout 900 -k "fused or bf16x3 or attention" > gpurun_out/r2_ops.log 2>&1; tail -15 gpurun_out/r2_ops.log
python tools/panel_bench.py attn > gpurun_out/r2_attn_fused.log 2>&1; cat gpurun_out/r2_attn_fused.log
python -m pytest tests -m gpu -q --maxfail 12 --timeout 900 > gpurun_out/r2_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_gpu_tests.log
tail -30 gpurun_out/r2_gpu_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r2_bench_fused.log 2> gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench_fused.log
PARSEQ_NO_FUSED_ATTN=1 python bench.py --no-cpu-baseline --no-parity > gpurun_out/r2_bench_unfused_attn.log 2>/dev/null; cat gpurun_out/r2_bench_unfused_attn.log
