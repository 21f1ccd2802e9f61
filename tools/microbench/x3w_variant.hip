// The two-waves-per-SIMD bf16x3 one-launch encoder (encoder_blocks_x3w.h) alone, as a small shared library with the entry point of
// x3_variant.hip: tools/x3_variants.sh builds it under -D switches (X3W_AHEAD, X3W_PINGPONG, X3W_PRIO, X3W_PARK_TILES, X3W_PHASES ...)
// and tools/x3_variant_bench.py times the builds against each other and against the four-wave kernel's builds.
#include "x3w_lab.h"
#include <vector>
#ifndef X3W_PHASES
#define X3W_PHASES 3           // bit 0 the attention branch, bit 1 the MLP branch (anything but 3: a phase alone, for timing)
#endif
using namespace pq;
extern "C" int x3_variant_run(float* x, const float* master, const void* pack, long long master_elems, const unsigned* offsets, int depth, int M,
                              void* table_ws, float* scratch, const unsigned* tail_offsets, float* kmem, float* vmem, void* stream) {
    std::vector<EncBlockParams> host(depth);
    for (int i = 0; i < depth; ++i) {
        const unsigned* o = offsets + (size_t)i * 12;
        EncBlockParams& e = host[i];
        e.ln1_w = o[0]; e.ln1_b = o[1]; e.wqkv = o[2]; e.bqkv = o[3]; e.wproj = o[4]; e.bproj = o[5];
        e.ln2_w = o[6]; e.ln2_b = o[7]; e.w1 = o[8]; e.b1 = o[9]; e.w2 = o[10]; e.b2 = o[11];
    }
    if (hipMemcpy(table_ws, host.data(), host.size() * sizeof(EncBlockParams), hipMemcpyHostToDevice) != hipSuccess) return -1;
    x3::EncTailX3 et{0, 0, 0, 0, nullptr, nullptr, 12};
    if (kmem) { et.norm_w = tail_offsets[0]; et.norm_b = tail_offsets[1]; et.wkv = tail_offsets[2]; et.bkv = tail_offsets[3]; et.kmem = kmem; et.vmem = vmem; }
    return (int)x3w::launch_enc_blocks_x3w<384, X3W_PHASES>((hipStream_t)stream, x, pack, (size_t)master_elems * 4, master, reinterpret_cast<const EncBlockParams*>(table_ws),
                                                            depth, 1e-6f, M, scratch, et);
}
