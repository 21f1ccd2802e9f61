mkdir -p gpurun_out
for i in 1 2; do timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_ops.py -m gpu -q --timeout 600 -k "x3 or distinct" 2>&1 | tail -2; done
for v in "" _nopre ""; do
  PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$v.so timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity --steps 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=d['kernel_families']; print('lib$v', d['value'], d['sequential_value'], {k:(v['avg_us'],v['launches_per_step']) for k,v in f.items() if k.startswith('enc')})"
done 2>&1 | tee gpurun_out/r2_x3_variants.log
