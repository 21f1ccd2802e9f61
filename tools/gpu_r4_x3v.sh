# Round 4: builds of the bf16x3 one-launch encoder against each other (tools/x3_variants.sh -> parseq_amd/lib/x3v/*.so), kernel alone
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
X3_ROUNDS=7 timeout 600 python tools/x3_variant_bench.py "$@" 2>&1 | tee gpurun_out/x3_variants.log | tail -12
