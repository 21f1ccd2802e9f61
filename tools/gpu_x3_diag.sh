# Runs tools/x3_diag2.py (bf16x3 LayerNorm-fused GEMM reproducer) on the product library and on PQ_DIAG_X3 builds (tools/enc_variant.sh diag<N> -DPQ_DIAG_X3=<N>).
for v in "" _diag8; do echo "=== lib$v"; PARSEQ_HIP_LIB=$PWD/parseq_amd/lib/libparseq_hip$v.so timeout 300 python tools/x3_diag2.py 2>&1 | grep -v amdgpu.ids | tail -40; done
