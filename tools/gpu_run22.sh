mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r2_gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/r2_gpu_tests.log | tail -6
timeout 600 python bench.py > gpurun_out/r2_bench.log 2>gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench.log | cut -c1-300; python -c "
import json; d=json.load(open('gpurun_out/r2_bench.log')); print(d['value'], d['sequential_value'], d['exact_value'], d['parity'])"
