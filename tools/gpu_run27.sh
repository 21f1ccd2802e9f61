mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 600 -k "bf16 and not x3 or tail or batch1 or replicas or hub or test_step or config3 or repeated" 2>&1 | tail -4
for v in "" _prev "" _prev; do
  PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$v', d['value'], d['sequential_value'], d['kernel_families']['enc.blocks_fused']['avg_us'], d['roofline']['frac'])"
done
