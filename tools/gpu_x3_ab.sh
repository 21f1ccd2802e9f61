# bf16x3 A/B: the product library against libparseq_hip_<suffix>.so builds given as arguments (tools/enc_variant.sh); x3 tests first
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_ops.py -m gpu -q --timeout 600 -k "x3 or distinct or pairs" 2>&1 | tail -2
for v in "" "$@" ""; do
  s=${v:+_$v}
  PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$s.so timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity --steps 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=d['kernel_families']; print('lib$s', d['value'], d['sequential_value'], {k:v['avg_us'] for k,v in f.items() if k.startswith('enc')})"
done
