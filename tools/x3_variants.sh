#!/bin/bash
# Builds parseq_amd/lib/x3v/<name>.so for every "name:flags" argument (in parallel), e.g.
#   tools/x3_variants.sh "base:" "ahead3:-DX3_AHEAD=3" "nogelu:-DX3_ABLATE=1" "w8:" "w8_mlp:-DX3W_PHASES=2"
cd "$(dirname "$0")/.."
mkdir -p parseq_amd/lib/x3v
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  src=tools/microbench/x3_variant.hip; case $name in w8*) src=tools/microbench/x3w_variant.hip ;; esac      # names starting with w8: the two-waves-per-SIMD kernel (encoder_blocks_x3w.h)
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -Iparseq_amd/csrc -Itools/microbench $flags -o parseq_amd/lib/x3v/$name.so $src \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|ScratchSize|VGPRs Spill" | sed "s/.*remark: */$name: /" ) &
done
wait
ls -la parseq_amd/lib/x3v/
