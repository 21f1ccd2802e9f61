# Round 4: 24-bit K / V rows and the in-launch patch head of the bf16x3 mode — their tests, the A/B against the switches that turn them off
# (same box, same build), and the kernel trace of the exact mode.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_ops.py -m gpu -q --timeout 600 -k "kv_rows or patch_head or uint8 or slots_and or baseline_configs or forward_fp32 or x3 or encoder_memory or reference_idiom" > gpurun_out/x3_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/x3_tests.log | tail -25
Q="--steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag: value', d['value'], 'seq', d['sequential_value'])"; }
run base X=1
run no_kv24 PARSEQ_NO_KV24=1
run no_head PARSEQ_NO_FUSED_HEAD=1
run neither PARSEQ_NO_KV24=1 PARSEQ_NO_FUSED_HEAD=1
run base_again X=1
timeout 200 python bench.py $Q --streams 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams 3: value', d['value'], 'seq', d['sequential_value'])"
rm -rf gpurun_out/prof_x3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x3 -o x3 -- python bench.py --steps 5 --warmup 2 --streams 1 --repeats 1 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode > gpurun_out/x3_prof.log 2>&1
S=$(find gpurun_out/prof_x3 -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/r04_rocprof_kernel_stats_bf16x3_v1.md; head -16 gpurun_out/r04_rocprof_kernel_stats_bf16x3_v1.md
rm -rf gpurun_out/prof_x3
