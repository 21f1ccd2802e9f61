# Round-end evidence on one MI355X (through gpurun): rocprofv3 kernel stats of bench.py's command in both timed modes, PMC FETCH / WRITE /
# MFMA-busy passes (separate --pmc runs, kernel-trace only); summaries land in gpurun_out/ and are copied to profiles/ by hand.
# (The GPU test suite and the bench line itself: tools/gpu_suite.sh.)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r03}
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_x3 gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-train > gpurun_out/prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats_x3 -o p -- python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-train > gpurun_out/prof_stats_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o p -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile --no-train > gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o p -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile --no-train > gpurun_out/prof_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_mfma -o p -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile --no-train > gpurun_out/prof_mfma.log 2>&1
S=$(find gpurun_out/prof_stats -name "*results.db" | head -1); X=$(find gpurun_out/prof_stats_x3 -name "*results.db" | head -1)
F=$(find gpurun_out/prof_fetch -name "*results.db" | head -1); W=$(find gpurun_out/prof_write -name "*results.db" | head -1); M=$(find gpurun_out/prof_mfma -name "*results.db" | head -1)
python tools/rocprof_summary.py $S > gpurun_out/${R}_rocprof_kernel_stats_bf16_final.md; head -12 gpurun_out/${R}_rocprof_kernel_stats_bf16_final.md
python tools/rocprof_summary.py $X > gpurun_out/${R}_rocprof_kernel_stats_bf16x3_final.md; head -12 gpurun_out/${R}_rocprof_kernel_stats_bf16x3_final.md
python tools/pmc_summary.py $F $W --json gpurun_out/${R}_pmc_traffic.json > gpurun_out/${R}_pmc_hbm_traffic.md; head -8 gpurun_out/${R}_pmc_hbm_traffic.md
python tools/pmc_mfma_summary.py $M > gpurun_out/${R}_pmc_mfma_util.md; head -8 gpurun_out/${R}_pmc_mfma_util.md
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_x3 gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
