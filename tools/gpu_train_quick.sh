# Training GPU tests, tools/train_bench.py (bf16-operand mode; PARSEQ_TRAIN_F32_ATTN=1 = encoder attention back on the fp32 kernels, A/B)
# and a rocprofv3 kernel summary of the step.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_training.py -m gpu -q --timeout 600 > gpurun_out/train_tests.log 2>&1; echo "tests exit $?"; grep -v "^  File" gpurun_out/train_tests.log | tail -12
timeout 600 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench.json | cut -c1-300
PARSEQ_TRAIN_F32_ATTN=1 timeout 600 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench_f32attn.json | cut -c1-300
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/train_prof.log 2>&1
S=$(find gpurun_out/prof_train -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/train_step_rocprof.md; head -24 gpurun_out/train_step_rocprof.md
python tools/rocprof_by_grid.py $S mfma_bgemm 24 > gpurun_out/train_gemm_by_grid.md; cat gpurun_out/train_gemm_by_grid.md
rm -rf gpurun_out/prof_train
