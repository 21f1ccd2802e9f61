# Round 3: four-workgroups-per-CU forms of the all-bf16 training GEMMs — training GPU tests (incl. the bit-identity test against the fp32-in-memory
# path), tools/train_bench.py default vs PARSEQ_TRAIN_GEMM_W3=1 (the three-workgroup kernels), rocprofv3 kernel summary + GEMMs by grid.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_training.py -m gpu -q --timeout 600 -s > gpurun_out/train_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed|error|worst per-tensor|bf16-operand step|bf16 decoder backward" gpurun_out/train_tests.log | tail -14
timeout 300 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench_default.json | cut -c1-200
PARSEQ_TRAIN_GEMM_W3=1 timeout 300 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench_w3.json | cut -c1-200
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/train_prof.log 2>&1
S=$(find gpurun_out/prof_train -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/train_step_rocprof.md; head -14 gpurun_out/train_step_rocprof.md | cut -c1-150
python tools/rocprof_by_grid.py $S mfma_bgemm16 30 > gpurun_out/train_gemm_by_grid.md; cat gpurun_out/train_gemm_by_grid.md
rm -rf gpurun_out/prof_train
