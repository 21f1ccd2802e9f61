for v in "" _diag8; do echo "=== lib$v"; PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$v.so timeout 300 python tools/x3_diag2.py 2>&1 | grep -v amdgpu.ids | tail -40; done
