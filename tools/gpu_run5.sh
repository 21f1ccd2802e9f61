set -x
mkdir -p gpurun_out
python tools/x3_diag2.py > gpurun_out/r2_x3_diag2.log 2>&1; cat gpurun_out/r2_x3_diag2.log
AMD_SERIALIZE_KERNEL=3 python tools/dbg_decode.py parseq-tiny > gpurun_out/r2_dbg_decode.log 2>&1; tail -8 gpurun_out/r2_dbg_decode.log
python -m pytest tests/test_hip_ops.py -m gpu -q --timeout 900 -k "encoder_blocks or fused" > gpurun_out/r2_ops.log 2>&1; tail -15 gpurun_out/r2_ops.log
python bench.py --no-cpu-baseline --no-parity --steps 50 > gpurun_out/r2_bench_blocks.log 2> gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench_blocks.log
