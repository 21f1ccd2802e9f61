cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/dbg_shadow.py 32 2>&1 | tail -8
timeout 600 python -m pytest tests/test_training.py -m gpu -q --timeout 600 -k "shadow or rounding_budget or full_step" 2>&1 | tail -5
