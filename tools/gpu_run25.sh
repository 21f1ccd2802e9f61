mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r2_gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/r2_gpu_tests.log | tail -6
python tools/panel_bench.py attn 2>&1 | tail -6
