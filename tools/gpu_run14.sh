mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q --timeout 600 -k "blocks or fused_mlp" 2>&1 | tail -3
for v in "" _block _t3 "" _block; do
  PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 30 --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$v', d['value'], d['kernel_families']['enc.blocks_fused']['avg_us'])"
done 2>&1 | tee gpurun_out/r2_enc_variants.log
