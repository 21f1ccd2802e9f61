# Round-end check on one MI355X (through gpurun): smoke(), the whole GPU suite, bench.py with its defaults (the driver's command).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/final_gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/final_gpu_tests.log | tail -4
timeout 600 python bench.py > gpurun_out/final_bench.json 2>gpurun_out/final_bench.err; python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['sequential_value'], d['exact_value'], d['roofline']['frac'], d['kernel_families']['enc.blocks_fused']['avg_us'], d['train'])"
