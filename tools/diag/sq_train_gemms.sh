cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmct$i
  PARSEQ_TRAIN_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmct$i -o p -- python tools/train_bench.py --steps 1 --warmup 1 > gpurun_out/pmct$i.log 2>&1 || tail -3 gpurun_out/pmct$i.log
done
for k in "mfma_bgemm16_kernel" "mfma_bgemm16t_kernel" "mfma_bgemm_kernelILb1ELb1E"; do echo "== $k"; python tools/pmc_generic.py $k $(find gpurun_out/pmct[0-9]* -name "*results.db"); done > gpurun_out/r06_train_gemm_sq_counters.md
rm -rf gpurun_out/pmct[0-9]*
cat gpurun_out/r06_train_gemm_sq_counters.md | head -80
