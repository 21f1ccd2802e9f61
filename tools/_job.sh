cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(X3_ROUNDS=15 timeout 900 python tools/x3_variant_bench.py prod_w8 w8q w8q_fo 2>&1 | tail -5) | tee gpurun_out/x3w_fo.log
