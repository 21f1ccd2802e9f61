# Round 3: the bf16x3 one-launch encoder (encoder_blocks_x3.h) on one MI355X: its op test, the bf16x3 parity cases, bench.py in that precision
# (one-launch vs per-op A/B), and a rocprofv3 kernel trace of the mode.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "x3" --timeout 600 > gpurun_out/r3_x3_ops.log 2>&1; echo "ops exit $?"; grep -v "^  File" gpurun_out/r3_x3_ops.log | tail -15
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "bf16x3 or distinct or repeated" --timeout 600 > gpurun_out/r3_x3_parity.log 2>&1; echo "parity exit $?"; tail -8 gpurun_out/r3_x3_parity.log
timeout 300 python bench.py --precision bf16x3 --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r3_x3_bench.log 2>gpurun_out/r3_x3_bench.err; tail -3 gpurun_out/r3_x3_bench.err; cat gpurun_out/r3_x3_bench.log
PARSEQ_NO_FUSED_X3=1 timeout 300 python bench.py --precision bf16x3 --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r3_x3_bench_perop.log 2>/dev/null; cat gpurun_out/r3_x3_bench_perop.log
rm -rf gpurun_out/prof_x3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x3 -o x3 -- python bench.py --precision bf16x3 --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r3_x3_prof.log 2>&1
S=$(find gpurun_out/prof_x3 -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/r03_rocprof_kernel_stats_bf16x3.md; head -25 gpurun_out/r03_rocprof_kernel_stats_bf16x3.md
rm -rf gpurun_out/prof_x3
