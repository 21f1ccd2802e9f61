set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail 10 --timeout 900 > gpurun_out/r2_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_gpu_tests.log
tail -40 gpurun_out/r2_gpu_tests.log
python tools/panel_bench.py ab > gpurun_out/r2_mlp_ab.log 2>&1; cat gpurun_out/r2_mlp_ab.log
python bench.py > gpurun_out/r2_bench.log 2> gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench.log
python bench.py --precision bf16x3 --steps 20 --no-cpu-baseline --no-parity > gpurun_out/r2_bench_x3.log 2> gpurun_out/r2_bench_x3.err; tail -3 gpurun_out/r2_bench_x3.err; cat gpurun_out/r2_bench_x3.log
