// libparseq_hip.so — the one-launch bf16 encoder (encoder_blocks.h) and its two branch kernels, compiled on their own.
#define PQ_INSTANTIATE_ENC_BLOCKS
#include <hip/hip_runtime.h>
#include "common.h"
#include "encoder_blocks.h"
