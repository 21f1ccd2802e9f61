# Round 4: the new bench line (exact mode as the headline), then the PARSEQ_X3_SPLIT A/B (the bf16x3 encoder in n launches: shorter persistent
# workgroups) in flight and one at a time, on one box.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r4.json 2>gpurun_out/bench_r4.err; tail -3 gpurun_out/bench_r4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4.json'))
print({k:d.get(k) for k in ('value','sequential_value','dtype','natural_exit_value','natural_exit_sequential_value','value_at_tolerance','tolerance_met_by_timed_dtype')})
print('roofline', d.get('roofline')); print('parity', d.get('parity')); tm=d.get('throughput_mode',{}); print('throughput_mode', {k:tm.get(k) for k in ('value','sequential_value','parity_vs_headline_mode','tolerance_met','roofline','error')})
print('train', d.get('train',{}).get('value'))
PY
Q="--steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode"
for sp in 1 2 3 4 6 12; do
  for st in 2 3; do
    PARSEQ_X3_SPLIT=$sp timeout 200 python bench.py $Q --streams $st 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split $sp streams $st: value', d['value'], 'seq', d['sequential_value'])"
  done
done
for q in 2 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py $Q --streams 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hwq $q: value', d['value'], 'seq', d['sequential_value'])"
done
