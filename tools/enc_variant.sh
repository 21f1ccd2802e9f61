#!/bin/bash
# tools/enc_variant.sh <suffix> <hipcc -D flags...>: builds parseq_amd/lib/libparseq_hip_<suffix>.so with the given defines (A/B builds)
cd "$(dirname "$0")/.."
suffix=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed "$@" -o parseq_amd/lib/libparseq_hip_$suffix.so parseq_amd/csrc/parseq_hip.hip
