#!/bin/bash
# One entry point for everything this repository runs on an MI355X through gpurun:  gpurun --timeout N -- 'bash tools/gpu_run.sh <task> [args]'
# Results land in gpurun_out/ (scratch, merged back by gpurun); what is worth keeping is copied to profiles/ by hand.  R = round tag of the file names.
#   suite            smoke() + the whole GPU test suite + the default bench line (the driver's round-end commands)
#   tests <k-expr>   pytest -m gpu -k <expr>
#   bench [args]     bench.py with the given arguments, the JSON line to gpurun_out/bench_<R>.json, headline keys printed
#   ab <name=ENV=V,ENV=V> ...   short exact-mode bench runs under each environment (switch A/Bs on one box)
#   profiles         rocprofv3 kernel stats (exact mode and bf16 mode) + FETCH / WRITE / MFMA-busy counter passes (separate --pmc runs) -> summaries
#   seqprof          rocprofv3 kernel stats of bench.py --streams 1 (one forward at a time), exact mode and bf16 mode
#   sq <kernel-substring> [bench args]   SQ counter passes over bench.py --steps 1 for one kernel (tools/pmc_generic.py)
#   tcc              L2 hit / miss and memory-side request counters for the decoder kernels and the encoder
#   train            training tests, tools/train_bench.py, kernel stats of a step; `train pmc` adds the counter passes
#   configs          bench.py on the other BASELINE / model configurations
#   scale [N ...]    tools/scale_curve.py: bench.py --gpus N for N in 1 2 4 8 (or the given list) -> profiles/scale.json (needs a multi-GPU box; an N the box cannot serve is recorded as refused)
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r06}
task=$1; shift
QUICK="--steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode --no-config3 --no-latency"
headline() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ('value', 'sequential_value', 'dtype', 'natural_exit_value', 'tolerance_met_by_timed_dtype')})
print('roofline', d.get('roofline')); print('parity', d.get('parity'))
tm = d.get('throughput_mode') or {}
print('throughput_mode', {k: tm.get(k) for k in ('value', 'sequential_value', 'parity_vs_headline_mode', 'tolerance_met', 'error')}, (tm.get('roofline') or {}).get('frac'))
print('train', d.get('train'))
PY
}
db() { find "$1" -name "*results.db" | head -1; }
case $task in
suite)
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/gpu_tests.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/gpu_tests.log | tail -15
  timeout 900 python bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err; tail -2 gpurun_out/bench_$R.err; headline gpurun_out/bench_$R.json ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "$1" > gpurun_out/gpu_tests_k.log 2>&1; echo "pytest exit $?"; grep -v "^  File" gpurun_out/gpu_tests_k.log | tail -25 ;;
bench)
  timeout 900 python bench.py "$@" > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err; tail -2 gpurun_out/bench_$R.err; headline gpurun_out/bench_$R.json ;;
ab)
  for spec in "$@"; do
    name=${spec%%=*}; envs=$(echo "${spec#*=}" | tr ',' ' ')
    env $envs timeout 300 python bench.py $QUICK 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name: value', d['value'], 'seq', d['sequential_value'])"
  done ;;
profiles)
  P="--steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode --no-config3 --no-latency"
  for prec in bf16x3 bf16; do
    rm -rf gpurun_out/prof_$prec
    timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$prec -o p -- python bench.py --precision $prec $P > gpurun_out/prof_$prec.log 2>&1
    python tools/rocprof_summary.py $(db gpurun_out/prof_$prec) > gpurun_out/${R}_rocprof_kernel_stats_$prec.md; head -12 gpurun_out/${R}_rocprof_kernel_stats_$prec.md
    for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      t=$(echo $c | cut -c1-5); rm -rf gpurun_out/pmc_${prec}_$t
      timeout 600 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_${prec}_$t -o p -- python bench.py --precision $prec --steps 2 --warmup 1 --streams 1 ${P#--steps 5 --warmup 2} > gpurun_out/pmc_${prec}_$t.log 2>&1
    done
    python tools/pmc_summary.py $(db gpurun_out/pmc_${prec}_FETCH) $(db gpurun_out/pmc_${prec}_WRITE) --json gpurun_out/${R}_pmc_traffic_$prec.json > gpurun_out/${R}_pmc_hbm_traffic_$prec.md; head -8 gpurun_out/${R}_pmc_hbm_traffic_$prec.md
    python tools/pmc_mfma_summary.py $(db gpurun_out/pmc_${prec}_SQ_VA) > gpurun_out/${R}_pmc_mfma_util_$prec.md; head -6 gpurun_out/${R}_pmc_mfma_util_$prec.md
    rm -rf gpurun_out/prof_$prec gpurun_out/pmc_${prec}_*
  done ;;
seqprof)   # kernel stats of one forward at a time (--streams 1: the latency form of the AR step), both matrix-core modes
  P="--steps 5 --warmup 2 --repeats 1 --streams 1 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode --no-config3 --no-latency"
  for prec in bf16x3 bf16; do
    rm -rf gpurun_out/prof_$prec
    timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$prec -o p -- python bench.py --precision $prec $P > gpurun_out/prof_$prec.log 2>&1
    python tools/rocprof_summary.py $(db gpurun_out/prof_$prec) > gpurun_out/${R}_rocprof_kernel_stats_${prec}_one_at_a_time.md; head -14 gpurun_out/${R}_rocprof_kernel_stats_${prec}_one_at_a_time.md
    tail -1 gpurun_out/prof_$prec.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$prec', d['value'], d['ms_per_step'])"
    rm -rf gpurun_out/prof_$prec
  done ;;
sq)
  kern=$1; shift; i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1)); rm -rf gpurun_out/pmc$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc$i -o p -- python bench.py --steps 1 --warmup 1 --streams 1 --repeats 1 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode "$@" > gpurun_out/pmc$i.log 2>&1 || tail -3 gpurun_out/pmc$i.log
  done
  python tools/pmc_generic.py $kern $(find gpurun_out/pmc[0-9]* -name "*results.db") | tee gpurun_out/${R}_${kern}_sq_counters.md
  rm -rf gpurun_out/pmc[0-9]* ;;
tcc)
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum"; do
    i=$((i+1)); rm -rf gpurun_out/pmcc$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcc$i -o p -- python bench.py --steps 1 --warmup 1 --streams 1 --repeats 1 --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode "$@" > gpurun_out/pmcc$i.log 2>&1 || tail -3 gpurun_out/pmcc$i.log
  done
  for k in dec_cross_attn_ar enc_blocks dec_step_mid dec_step_mlp; do echo "== $k"; python tools/pmc_generic.py $k $(find gpurun_out/pmcc* -name "*results.db"); done | tee gpurun_out/${R}_tcc_counters.md
  rm -rf gpurun_out/pmcc[0-9]* ;;
train)
  timeout 1200 python -m pytest tests/test_training.py -m gpu -q --timeout 900 2>&1 | tail -3
  timeout 600 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/${R}_train_bench.json | cut -c1-300
  rm -rf gpurun_out/prof_train
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 2 --warmup 1 > gpurun_out/train_prof.log 2>&1
  python tools/rocprof_summary.py $(db gpurun_out/prof_train) > gpurun_out/${R}_train_step_rocprof.md; head -16 gpurun_out/${R}_train_step_rocprof.md | cut -c1-170
  python tools/rocprof_by_grid.py $(db gpurun_out/prof_train) mfma_bgemm > gpurun_out/${R}_train_gemm_by_grid.md 2>/dev/null; head -12 gpurun_out/${R}_train_gemm_by_grid.md | cut -c1-170
  rm -rf gpurun_out/prof_train
  if [ "$1" = pmc ]; then
    for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
      t=$(echo $c | cut -c1-5); rm -rf gpurun_out/tp_$t
      timeout 240 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/tp_$t -o p -- python tools/train_bench.py --steps 1 --warmup 1 > gpurun_out/tp_$t.log 2>&1 || tail -3 gpurun_out/tp_$t.log
    done
    python tools/pmc_mfma_summary.py $(db gpurun_out/tp_SQ_VA) > gpurun_out/${R}_train_pmc_mfma_util.md; head -12 gpurun_out/${R}_train_pmc_mfma_util.md | cut -c1-200
    python tools/pmc_summary.py $(db gpurun_out/tp_FETCH) $(db gpurun_out/tp_WRITE) > gpurun_out/${R}_train_pmc_hbm_traffic.md; head -12 gpurun_out/${R}_train_pmc_hbm_traffic.md | cut -c1-200
    rm -rf gpurun_out/tp_*
  fi ;;
configs)
  run() { timeout 300 python bench.py --no-cpu-baseline --no-parity --no-profile --no-train --no-natural-exit --no-throughput-mode --no-config3 --no-latency --steps 40 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['value'], d['sequential_value'])"; }
  run --batch 1024 --refine-iters 2; run --natural-exit; run --precision bf16; run --precision bf16 --batch 1024 --refine-iters 2
  run --precision fp32 --batch 128; run --model parseq-tiny; run --model vitstr --precision bf16; run --model parseq-patch16-224 --batch 64 --precision bf16
  run --batch 256; run --batch 128; run --model vitstr; run --model parseq-patch16-224 --batch 64; run --model parseq-tiny --precision bf16 ;;
scale)
  if [ $# -gt 0 ]; then G="--gpus $*"; else G=""; fi
  timeout 3000 python tools/scale_curve.py $G --out gpurun_out/scale.json -- --steps 40 --repeats 3 --no-cpu-baseline --no-parity --no-profile --no-natural-exit --no-throughput-mode --no-config3 | tail -1 | cut -c1-1500 ;;
*) echo "unknown task $task"; exit 2 ;;
esac
