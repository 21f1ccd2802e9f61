cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(X3_ROUNDS=9 timeout 600 python tools/x3_variant_bench.py base w8p w8p_p4 w8p_p5 2>&1 | tail -6;  X3_ROUNDS=3 timeout 300 python tools/x3_variant_bench.py --timers8 w8p_p4_t 2>&1 | tail -38) | tee gpurun_out/x3w_prio_toggle.log
