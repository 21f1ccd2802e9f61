set -x
mkdir -p gpurun_out
AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=1 python -X faulthandler -m pytest tests/test_hip_parity.py -m gpu -q -x -s --timeout 600 -k "test_decode_and_head_reference_idiom" > gpurun_out/r2_decode_dbg.log 2>&1; echo "exit $?" >> gpurun_out/r2_decode_dbg.log
grep -v "^  File\|pluggy\|_pytest" gpurun_out/r2_decode_dbg.log | tail -40
python -m pytest tests -m gpu -q --maxfail 12 --timeout 900 --deselect tests/test_hip_parity.py::test_decode_and_head_reference_idiom > gpurun_out/r2_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_gpu_tests.log
grep -v "^  File" gpurun_out/r2_gpu_tests.log | tail -40
python bench.py > gpurun_out/r2_bench.log 2> gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench.log
