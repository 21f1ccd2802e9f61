// libparseq_hip.so — the one-launch bf16x3 encoder on two waves per SIMD (encoder_blocks_x3w.h), compiled on its own WITH -mllvm -amdgpu-mfma-vgpr-form
// (parseq_amd/build.py UNIT_FLAGS): the accumulators of this unit live in VGPRs.
#define PQ_INSTANTIATE_ENC_BLOCKS_X3W
#include <hip/hip_runtime.h>
#include "common.h"
#include "encoder_blocks.h"
#include "encoder_blocks_x3w.h"
