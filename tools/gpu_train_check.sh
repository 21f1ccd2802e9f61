# Training step on one MI355X (through gpurun): the training GPU tests, tools/train_bench.py — once per "NAME=VALUE" argument as an A/B against the
# default (e.g. tools/gpu_train_check.sh PARSEQ_TRAIN_PERM_GROUP=1 PARSEQ_TRAIN_GEMM_W3=1 PARSEQ_TRAIN_ATTN_KT8=1 PARSEQ_TRAIN_NO_PASS_LOOP=1) —
# and a rocprofv3 kernel summary of the default + the GEMM launches by grid shape.  This is the script behind profiles/r03_train_*_v8 .. v11.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_training.py -m gpu -q --timeout 600 -s > gpurun_out/train_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed|error|worst per-tensor|bf16-operand step|bf16 decoder backward" gpurun_out/train_tests.log | tail -14
timeout 300 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench_default.json | cut -c1-200
for kv in "$@"; do env "$kv" timeout 300 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee "gpurun_out/train_bench_${kv%%=*}.json" | cut -c1-200; done
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/train_prof.log 2>&1
S=$(find gpurun_out/prof_train -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/train_step_rocprof.md; head -30 gpurun_out/train_step_rocprof.md | cut -c1-150
python tools/rocprof_by_grid.py $S mfma_bgemm 30 > gpurun_out/train_gemm_by_grid.md
rm -rf gpurun_out/prof_train
