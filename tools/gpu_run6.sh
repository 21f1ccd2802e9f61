set -x
mkdir -p gpurun_out
python tools/x3_diag2.py > gpurun_out/r2_x3_diag2.log 2>&1; grep "bf16x3  M=4096\|bf16x3  M=13312" gpurun_out/r2_x3_diag2.log
PARSEQ_GEMM_EXTRA_LDS=16384 python tools/x3_diag2.py > gpurun_out/r2_x3_diag2_1wg.log 2>&1; grep "bf16x3  M=4096\|bf16x3  M=13312" gpurun_out/r2_x3_diag2_1wg.log
python -m pytest tests/test_hip_ops.py -m gpu -q --timeout 900 -k "encoder_blocks or fused" > gpurun_out/r2_ops.log 2>&1; tail -8 gpurun_out/r2_ops.log
python tools/x3_diag.py 2>&1 | grep "decoder only\|repeat" 
python bench.py --no-cpu-baseline --steps 50 > gpurun_out/r2_bench_blocks.log 2> gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench_blocks.log
