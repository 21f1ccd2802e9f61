#!/bin/bash
# Builds timing-ablation variants of the one-launch encoder (EB_ABLATE bits, encoder_blocks.h) next to the product library.
# Results of these builds are wrong by construction; they only answer "what does this part of the kernel cost".
cd "$(dirname "$0")/.."
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -DEB_ABLATE=$v -o parseq_amd/lib/libparseq_hip_abl$v.so parseq_amd/csrc/parseq_hip.hip &
done
wait
