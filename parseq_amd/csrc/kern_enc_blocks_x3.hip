// libparseq_hip.so — the one-launch bf16x3 encoder (encoder_blocks_x3.h), compiled on its own.
#define PQ_INSTANTIATE_ENC_BLOCKS_X3
#include <hip/hip_runtime.h>
#include "common.h"
#include "encoder_blocks.h"
#include "encoder_blocks_x3.h"
