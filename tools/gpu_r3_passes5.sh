# Round 3, last build: training GPU tests + tools/train_bench.py + kernel summary.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_training.py -m gpu -q --timeout 500 > gpurun_out/train_tests.log 2>&1; echo "tests exit $?"; tail -2 gpurun_out/train_tests.log
timeout 200 python tools/train_bench.py --steps 5 --warmup 2 2>/dev/null | tee gpurun_out/train_bench_default.json | cut -c1-200
rm -rf gpurun_out/prof_train
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/train_prof.log 2>&1
S=$(find gpurun_out/prof_train -name "*results.db" | head -1); python tools/rocprof_summary.py $S > gpurun_out/train_step_rocprof.md; head -9 gpurun_out/train_step_rocprof.md | cut -c1-150; tail -1 gpurun_out/train_step_rocprof.md
rm -rf gpurun_out/prof_train
