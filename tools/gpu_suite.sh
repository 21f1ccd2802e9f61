# Whole GPU test suite + smoke + the default bench line (with its training leg) on one MI355X (through gpurun).
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/gpu_tests.log | tail -15
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -2 gpurun_out/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench.log')); print('value', d['value'], 'seq', d['sequential_value'], 'exact', d['exact_value'], d.get('exact_sequential_value'), 'frac', d['roofline']['frac'], 'train', d.get('train'))"
timeout 600 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench.json | cut -c1-200
