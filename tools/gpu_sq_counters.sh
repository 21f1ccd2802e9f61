# SQ counter passes (separate rocprofv3 --pmc runs) over the one-launch encoder: profiles/r02_enc_blocks_sq_counters.md.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/r2_sq_counters.txt; wc -w gpurun_out/r2_sq_counters.txt
export PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip_t3.so
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc$i -o p -- python bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/pmc$i.log 2>&1 || tail -3 gpurun_out/pmc$i.log
done
python tools/pmc_generic.py enc_blocks $(find gpurun_out/pmc* -name "*results.db") | tee gpurun_out/r2_enc_blocks_sq.md
rm -rf gpurun_out/pmc[0-9]*
