# Round 3 check on one MI355X: bf16x3 op + parity tests, bench.py (default), decoder-priority A/B (bf16 and bf16x3), short.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_parity.py -m gpu -q -x -k "x3 or distinct or repeated or slots" --timeout 600 > gpurun_out/r3_check_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/r3_check_tests.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench.log 2>gpurun_out/r3_bench.err; tail -3 gpurun_out/r3_bench.err; cat gpurun_out/r3_bench.log
for prec in bf16 bf16x3; do
  timeout 200 python bench.py --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r3_prio0_$prec.log 2>/dev/null
  PARSEQ_DEC_PRIORITY=1 timeout 200 python bench.py --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-profile > gpurun_out/r3_prio1_$prec.log 2>/dev/null
  python - <<PY
import json
for tag in ('prio0', 'prio1'):
    try:
        r = json.loads(open('gpurun_out/r3_%s_$prec.log' % tag).read().strip().splitlines()[-1])
        print('$prec', tag, 'value', r['value'], 'sequential', r['sequential_value'], r['repeats'])
    except Exception as e:
        print('$prec', tag, 'failed', e)
PY
done
