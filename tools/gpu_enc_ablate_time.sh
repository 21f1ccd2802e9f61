# Times the EB_ABLATE builds made by tools/enc_ablate.sh (libparseq_hip_abl<mask>.so) with bench.py: profiles/r02_enc_ablation.log.
mkdir -p gpurun_out
for v in "" _abl1 _abl2 _abl4 _abl8 _abl16 _abl28 _abl31 ""; do
  PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 20 --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$v', d['value'], d['kernel_families']['enc.blocks_fused']['avg_us'])"
done 2>&1 | tee gpurun_out/r2_enc_ablate.log
