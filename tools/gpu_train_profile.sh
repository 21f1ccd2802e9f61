# Training step (row N3): tools/train_bench.py in both GEMM modes and the rocprofv3 --kernel-trace --stats summary of the bf16-operand mode.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py -m gpu -q --timeout 600 -k "bf16_operand_step or full_step or batch_64" -s 2>&1 | grep -E "bf16-operand step|passed|failed|Error" | tail -5; timeout 900 python -m pytest tests/test_training.py -m gpu -q --timeout 600 2>&1 | tail -2
for tp in fp32 bf16; do timeout 600 python tools/train_bench.py --steps 3 --warmup 1 --train-precision $tp 2>/dev/null | tee gpurun_out/r02_train_bench_$tp.json | cut -c1-260; done
rm -rf gpurun_out/prof_train
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o t -- python tools/train_bench.py --steps 2 --warmup 1 > gpurun_out/r02_train_prof.log 2>&1
T=$(find gpurun_out/prof_train -name "*results.db" | head -1)
python tools/rocprof_summary.py $T > gpurun_out/r02_train_step_rocprof_bf16.md; head -12 gpurun_out/r02_train_step_rocprof_bf16.md | cut -c1-160
rm -rf gpurun_out/prof_train
