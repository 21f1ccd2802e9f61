# SQ / TCC counter passes (separate rocprofv3 --pmc runs) over the training step's GEMM kernel.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/tpmc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/tpmc$i -o p -- python tools/train_bench.py --steps 1 --warmup 1 > gpurun_out/tpmc$i.log 2>&1 || tail -3 gpurun_out/tpmc$i.log
done
python tools/pmc_generic.py mfma_bgemm $(find gpurun_out/tpmc* -name "*results.db") | tee gpurun_out/r03_train_gemm_counters.md
python tools/pmc_generic.py colsum $(find gpurun_out/tpmc* -name "*results.db") | tee -a gpurun_out/r03_train_gemm_counters.md
rm -rf gpurun_out/tpmc[0-9]*
