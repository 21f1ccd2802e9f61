# Round 3, fused AR step on bf16 pairs (decoder_step.h X3) + refinement-pass kernels: whole GPU suite, bench.py (default line), the
# exact-tolerance mode with 2 / 3 batches in flight and through the per-op AR step (A/B), rocprofv3 kernel trace + MFMA-busy of the mode.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r3_gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/r3_gpu_tests.log | tail -25
timeout 500 python bench.py > gpurun_out/r3_bench.log 2>gpurun_out/r3_bench.err; tail -2 gpurun_out/r3_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench.log')); print('value', d['value'], 'seq', d['sequential_value'], 'exact', d['exact_value'], d.get('exact_sequential_value'), 'frac', d['roofline']['frac'], d['parity'])"
for st in 2 3; do
  timeout 200 python bench.py --precision bf16x3 --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/r3_x3_streams$st.log 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r3_x3_streams$st.log')); print('x3 streams $st value', d['value'], 'seq', d['sequential_value'], {k: (v['avg_us'], v['launches_per_step']) for k, v in d['kernel_families'].items()})"
done
PARSEQ_NO_FUSED_STEP=1 timeout 200 python bench.py --precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r3_x3_perop_step.log 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r3_x3_perop_step.log')); print('x3 per-op AR step value', d['value'], 'seq', d['sequential_value'])"
rm -rf gpurun_out/prof_x3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x3 -o x3 -- python bench.py --precision bf16x3 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r3_x3_prof.log 2>&1
S=$(find gpurun_out/prof_x3 -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/r03_rocprof_kernel_stats_bf16x3_v2.md; head -30 gpurun_out/r03_rocprof_kernel_stats_bf16x3_v2.md
rm -rf gpurun_out/prof_x3 gpurun_out/pmc_x3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_x3 -o p -- python bench.py --precision bf16x3 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r3_x3_pmc.log 2>&1
S=$(find gpurun_out/pmc_x3 -name "*results.db" | head -1); python tools/pmc_mfma_summary.py $S > gpurun_out/r03_pmc_mfma_util_bf16x3.md; head -12 gpurun_out/r03_pmc_mfma_util_bf16x3.md
rm -rf gpurun_out/pmc_x3
