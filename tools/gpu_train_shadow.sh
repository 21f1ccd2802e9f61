# Training GPU tests, tools/train_bench.py with bf16 shadow operands (default) and with PARSEQ_TRAIN_NO_SHADOWS=1 (A/B), rocprof of the step
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_training.py -m gpu -q --timeout 600 > gpurun_out/train_tests.log 2>&1; echo "tests exit $?"; grep -v "^  File" gpurun_out/train_tests.log | tail -12
timeout 600 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench.json | cut -c1-300
PARSEQ_TRAIN_SHADOW_LEVEL=1 timeout 600 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench_level1.json | cut -c1-300
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/train_prof.log 2>&1
S=$(find gpurun_out/prof_train -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/train_step_rocprof.md; head -24 gpurun_out/train_step_rocprof.md
python tools/rocprof_by_grid.py $S mfma_bgemm 30 > gpurun_out/train_gemm_by_grid.md; cat gpurun_out/train_gemm_by_grid.md
rm -rf gpurun_out/prof_train
