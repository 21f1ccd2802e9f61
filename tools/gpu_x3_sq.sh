# SQ counter passes (separate rocprofv3 --pmc runs) over the bf16x3 one-launch encoder (tools/x3_variant_bench.py, build "base")
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc$i
  X3_ROUNDS=2 timeout 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc$i -o p -- python tools/x3_variant_bench.py ${X3_BUILD:-base} > gpurun_out/pmc$i.log 2>&1 || tail -3 gpurun_out/pmc$i.log
done
python tools/pmc_generic.py enc_blocks_x3 $(find gpurun_out/pmc* -name "*results.db") | tee gpurun_out/r03_x3_enc_blocks_sq.md
rm -rf gpurun_out/pmc[0-9]*
