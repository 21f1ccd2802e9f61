// gfx950: does a VALU write of a 128-bit buffer store's data registers, issued right behind the store, reach memory?
// LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) treats MUBUF stores of more than 64 bits as hazardous only when
// soffset is NOT a register; encoder_blocks.h's record stores (SGPR soffset) showed dword 1 of lanes 12-15 of every row of 16 replaced
// by what the next v_pk_add_f32 wrote into the data registers.  Modes: soffset in an SGPR / soffset = 0; the clobbering instruction
// v_pk_add_f32 / v_mov_b32; 0..3 s_nop between the store and the clobber.
//   hipcc --offload-arch=gfx950 -O3 -o store_hazard store_hazard.hip && ./store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define STORE_SGPR "buffer_store_dwordx4 v[20:23], v24, %[r], %[soff] offen\n\t"
#define STORE_ZERO "buffer_store_dwordx4 v[20:23], v24, %[r], 0 offen\n\t"
#define CLOB_PK "v_pk_add_f32 v[20:21], v[26:27], v[26:27]\n\tv_pk_add_f32 v[22:23], v[26:27], v[26:27]\n\t"
#define CLOB_MOV "v_mov_b32 v20, v26\n\tv_mov_b32 v21, v26\n\tv_mov_b32 v22, v26\n\tv_mov_b32 v23, v26\n\t"
#define BODY(STORE, NOPS, CLOB)                                                                                                    \
    asm volatile("v_mov_b32 v20, %[d0]\n\tv_mov_b32 v21, %[d1]\n\tv_mov_b32 v22, %[d2]\n\tv_mov_b32 v23, %[d3]\n\t"               \
                 "v_mov_b32 v24, %[voff]\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\ts_nop 7\n\t" STORE NOPS CLOB                  \
                 "s_waitcnt vmcnt(0)\n\t" ::[d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [voff] "v"(voff), [r] "s"(r), \
                 [soff] "s"(soff)                                                                                                 \
                 : "v20", "v21", "v22", "v23", "v24", "v26", "v27", "memory")

template <int MODE>
__global__ void k(unsigned* out, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1u << 30, 0x00020000);
    const unsigned lane = threadIdx.x;
    const unsigned d0 = 0xA0000000u | lane, d1 = 0xA1000000u | lane, d2 = 0xA2000000u | lane, d3 = 0xA3000000u | lane;
    const unsigned voff = lane * 16 + blockIdx.x * 1024;
    if constexpr (MODE == 0) BODY(STORE_SGPR, "", CLOB_PK);
    if constexpr (MODE == 1) BODY(STORE_SGPR, "s_nop 0\n\t", CLOB_PK);
    if constexpr (MODE == 2) BODY(STORE_SGPR, "s_nop 1\n\t", CLOB_PK);
    if constexpr (MODE == 3) BODY(STORE_SGPR, "s_nop 2\n\t", CLOB_PK);
    if constexpr (MODE == 4) BODY(STORE_SGPR, "", CLOB_MOV);
    if constexpr (MODE == 5) BODY(STORE_SGPR, "s_nop 0\n\t", CLOB_MOV);
    if constexpr (MODE == 6) BODY(STORE_ZERO, "", CLOB_PK);
    if constexpr (MODE == 7) BODY(STORE_ZERO, "s_nop 0\n\t", CLOB_PK);
    if constexpr (MODE == 8) BODY(STORE_ZERO, "", CLOB_MOV);
    if constexpr (MODE == 9) BODY(STORE_ZERO, "s_nop 1\n\t", CLOB_PK);
}

template <int MODE>
void run(const char* what) {
    const int blocks = 4096;
    unsigned* dev;
    hipMalloc(&dev, (size_t)blocks * 1024 + 4096);
    hipMemset(dev, 0, (size_t)blocks * 1024 + 4096);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, dev, 64u);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)blocks * 256 + 16);
    hipMemcpy(h.data(), dev, h.size() * 4, hipMemcpyDeviceToHost);
    const int so = MODE < 6 ? 16 : 0;      // soff = 64 bytes = 16 dwords in the SGPR modes
    long bad = 0, per_dword[4] = {0, 0, 0, 0}, per_lane4[16] = {0};
    for (int b = 0; b < blocks; ++b)
        for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
                const unsigned want = (0xA0000000u + (d << 24)) | l, got = h[(size_t)b * 256 + so + l * 4 + d];
                if (got != want) { ++bad; ++per_dword[d]; ++per_lane4[(l & 15)]; }
            }
    printf("%-52s bad %8ld of %d   by dword %ld %ld %ld %ld   by lane%%16:", what, bad, blocks * 256, per_dword[0], per_dword[1], per_dword[2], per_dword[3]);
    for (int i = 0; i < 16; ++i) printf(" %ld", per_lane4[i]);
    printf("\n");
    hipFree(dev);
}
int main() {
    run<0>("SGPR soffset, v_pk_add_f32 right behind");
    run<1>("SGPR soffset, s_nop 0, v_pk_add_f32");
    run<2>("SGPR soffset, s_nop 1, v_pk_add_f32");
    run<3>("SGPR soffset, s_nop 2, v_pk_add_f32");
    run<4>("SGPR soffset, v_mov_b32 right behind");
    run<5>("SGPR soffset, s_nop 0, v_mov_b32");
    run<6>("soffset 0, v_pk_add_f32 right behind");
    run<7>("soffset 0, s_nop 0, v_pk_add_f32");
    run<8>("soffset 0, v_mov_b32 right behind");
    run<9>("soffset 0, s_nop 1, v_pk_add_f32");
    return 0;
}
