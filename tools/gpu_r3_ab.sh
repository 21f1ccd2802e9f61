# Round 3 A/B on one MI355X: the op tests fixed after the first run, then cross-attention K / V loads non-temporal vs default policy
# (PARSEQ_CA_TEMPORAL=1) in both timed modes.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_parity.py -m gpu -q -k "patch_head or kv_tail or batch_520 or memory_cache" --timeout 600 > gpurun_out/r3_ab_tests.log 2>&1; echo "tests exit $?"; grep -v "^  File" gpurun_out/r3_ab_tests.log | tail -12
for prec in bf16 bf16x3; do
  for t in 0 1; do
    if [ $t = 1 ]; then export PARSEQ_CA_TEMPORAL=1; else unset PARSEQ_CA_TEMPORAL; fi
    timeout 200 python bench.py --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/r3_ca_t${t}_$prec.log 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/r3_ca_t${t}_$prec.log')); print('$prec temporal=$t value', d['value'], 'seq', d['sequential_value'], {k: v['avg_us'] for k, v in d['kernel_families'].items() if k.startswith('dec.')})"
  done
done
