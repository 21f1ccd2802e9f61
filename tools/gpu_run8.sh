set -x
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 900 -k "chains or replicas or slots or repeated or batch_520 or ragged or forward_bf16 or fp32_matches" > gpurun_out/r2_chain_tests.log 2>&1; grep -v "^  File" gpurun_out/r2_chain_tests.log | tail -15
for c in 1 2 4 8; do PARSEQ_AR_CHAINS=$c python bench.py --no-cpu-baseline --no-parity --no-profile --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chains $c', d['value'], d['sequential_value'])"; done
python bench.py --no-cpu-baseline --no-parity --steps 50 > gpurun_out/r2_bench_chains.log 2>gpurun_out/r2_bench.err; cat gpurun_out/r2_bench_chains.log | cut -c1-400
