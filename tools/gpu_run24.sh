mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q --timeout 600 -k "pairs" 2>&1 | tail -3
for v in "" _ring12 _ring16 ""; do
  PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 50 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=d['kernel_families']; print('lib$v', d['value'], d['sequential_value'], {k:v['avg_us'] for k,v in f.items() if k.startswith('dec.step') or k=='dec.cross_attention'})"
done
PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip_ring16.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 600 -k "bf16 and not x3" 2>&1 | tail -2
