# Round 4, first box: smoke + the GPU suite + the default bench line on the seven-unit build, then the counter passes VERDICT r3 asked for on
# the bf16x3 one-launch encoder (FETCH_SIZE / WRITE_SIZE / MFMA-busy, separate --pmc runs, kernel-trace only).
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/gpu_tests.log | tail -15
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -2 gpurun_out/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench.log')); print('value', d['value'], 'seq', d['sequential_value'], 'exact', d['exact_value'], d.get('exact_sequential_value'), 'frac', d['roofline']['frac'], 'train', d.get('train'))"
R=r04
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
X="python bench.py --precision bf16x3 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile --no-train"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o p -- $X > gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o p -- $X > gpurun_out/prof_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_mfma -o p -- $X > gpurun_out/prof_mfma.log 2>&1
F=$(find gpurun_out/prof_fetch -name "*results.db" | head -1); W=$(find gpurun_out/prof_write -name "*results.db" | head -1); M=$(find gpurun_out/prof_mfma -name "*results.db" | head -1)
python tools/pmc_summary.py $F $W --json gpurun_out/${R}_pmc_traffic_bf16x3.json > gpurun_out/${R}_pmc_hbm_traffic_bf16x3.md; head -12 gpurun_out/${R}_pmc_hbm_traffic_bf16x3.md
python tools/pmc_mfma_summary.py $M > gpurun_out/${R}_pmc_mfma_util_bf16x3.md; head -8 gpurun_out/${R}_pmc_mfma_util_bf16x3.md
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
