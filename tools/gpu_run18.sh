timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_ops.py -m gpu -q --timeout 600 -k "x3 or distinct" 2>&1 | grep -E "^\[|FAILED|passed|failed|Error|assert " | tail -30
echo ---- lnsep
PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip_lnsep.so timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 600 -k "distinct" 2>&1 | grep -E "^\[|FAILED|passed|failed|Error|assert " | tail -12
