# Round-end validation on one MI355X (through gpurun): full GPU test suite, bench.py, rocprofv3 kernel stats, PMC FETCH / WRITE / MFMA-busy passes; summaries land in gpurun_out/ and are copied to profiles/ by hand.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r2_gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/r2_gpu_tests.log | tail -8
timeout 600 python bench.py > gpurun_out/r2_bench.log 2>gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; cat gpurun_out/r2_bench.log
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o r2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/r2_prof_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o r2 -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r2_prof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o r2 -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r2_prof_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_mfma -o r2 -- python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r2_prof_mfma.log 2>&1
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma -name "*.db" | head
S=$(find gpurun_out/prof_stats -name "*results.db" | head -1); F=$(find gpurun_out/prof_fetch -name "*results.db" | head -1); W=$(find gpurun_out/prof_write -name "*results.db" | head -1); M=$(find gpurun_out/prof_mfma -name "*results.db" | head -1)
python tools/rocprof_summary.py $S > gpurun_out/r02_rocprof_kernel_stats.md; head -12 gpurun_out/r02_rocprof_kernel_stats.md
python tools/pmc_summary.py $F $W --json gpurun_out/r02_pmc_traffic.json > gpurun_out/r02_pmc_hbm_traffic.md; head -8 gpurun_out/r02_pmc_hbm_traffic.md
python tools/pmc_mfma_summary.py $M > gpurun_out/r02_pmc_mfma_util.md; head -8 gpurun_out/r02_pmc_mfma_util.md
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
