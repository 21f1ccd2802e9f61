#!/bin/bash
# Builds parseq_amd/lib/x3v/<name>.so for every "name:flags" argument (in parallel), e.g.
#   tools/x3_variants.sh "base:" "ahead3:-DX3_AHEAD=3" "nogelu:-DX3_ABLATE=1"
cd "$(dirname "$0")/.."
mkdir -p parseq_amd/lib/x3v
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -Iparseq_amd/csrc $flags -o parseq_amd/lib/x3v/$name.so tools/microbench/x3_variant.hip \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|ScratchSize|VGPRs Spill" | sed "s/.*remark: */$name: /" ) &
done
wait
ls -la parseq_amd/lib/x3v/
