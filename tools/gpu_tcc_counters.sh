# L2 hit / miss and memory-side read-request counters for the decoder kernels and the encoder: profiles/r02_tcc_l2_counters.md.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmcc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcc$i -o p -- python bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-parity --no-profile > gpurun_out/pmcc$i.log 2>&1 || tail -3 gpurun_out/pmcc$i.log
done
for k in dec_cross_attn_ar enc_blocks dec_step_mid dec_step_mlp; do echo "== $k"; python tools/pmc_generic.py $k $(find gpurun_out/pmcc* -name "*results.db"); done | tee gpurun_out/r2_tcc_counters.md
rm -rf gpurun_out/pmcc[0-9]*
