# Training step (row N3): tools/train_bench.py line and its rocprofv3 --kernel-trace --stats summary (profiles/r02_train_*).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/r02_train_bench.json 2> gpurun_out/r02_train_bench.err; tail -2 gpurun_out/r02_train_bench.err; cat gpurun_out/r02_train_bench.json | cut -c1-400
rm -rf gpurun_out/prof_train
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 2 --warmup 1 > gpurun_out/r02_train_prof.log 2>&1
T=$(find gpurun_out/prof_train -name "*results.db" | head -1)
python tools/rocprof_summary.py $T > gpurun_out/r02_train_step_rocprof.md; head -16 gpurun_out/r02_train_step_rocprof.md | cut -c1-200
rm -rf gpurun_out/prof_train
