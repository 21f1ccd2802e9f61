# Training step, counter evidence on the round's last build: MFMA-busy and HBM FETCH / WRITE per kernel (three separate rocprofv3 --pmc passes,
# kernel-trace only) over tools/train_bench.py --steps 1 --warmup 1.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tp_mfma gpurun_out/tp_fetch gpurun_out/tp_write
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/tp_mfma -o p -- python tools/train_bench.py --steps 1 --warmup 1 > gpurun_out/tp_mfma.log 2>&1 || tail -3 gpurun_out/tp_mfma.log
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/tp_fetch -o p -- python tools/train_bench.py --steps 1 --warmup 1 > gpurun_out/tp_fetch.log 2>&1 || tail -3 gpurun_out/tp_fetch.log
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/tp_write -o p -- python tools/train_bench.py --steps 1 --warmup 1 > gpurun_out/tp_write.log 2>&1 || tail -3 gpurun_out/tp_write.log
M=$(find gpurun_out/tp_mfma -name "*results.db" | head -1); F=$(find gpurun_out/tp_fetch -name "*results.db" | head -1); W=$(find gpurun_out/tp_write -name "*results.db" | head -1)
python tools/pmc_mfma_summary.py $M > gpurun_out/r03_train_pmc_mfma_util.md; head -12 gpurun_out/r03_train_pmc_mfma_util.md | cut -c1-200
python tools/pmc_summary.py $F $W > gpurun_out/r03_train_pmc_hbm_traffic.md; head -12 gpurun_out/r03_train_pmc_hbm_traffic.md | cut -c1-200
rm -rf gpurun_out/tp_mfma gpurun_out/tp_fetch gpurun_out/tp_write
