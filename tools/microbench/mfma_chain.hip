// How fast does ONE wave per SIMD issue the bf16x3 product's MFMA pattern?  Registers only (no LDS, no memory): the three-product update of
// encoder_blocks_x3.h mma3_w — c0 += wl*ah0; c1 += wl*ah1; c0 += wh*al0; c1 += wh*al1; c0 += wh*ah0; c1 += wh*ah1 — revisits each accumulator
// every SECOND v_mfma_f32_16x16x32_bf16; variants spread the dependent updates further apart.  Prints cycles per MFMA (16 = the pipe's rate).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain tools/microbench/mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// volatile asm: the compiler would otherwise reorder independent MFMAs (it spreads dependent ones apart on its own) and the orders below would not be what runs
__device__ __forceinline__ f32x4 mfma_v(const bf16x8& a, const bf16x8& b, f32x4 c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
}
#define MFMA(a, b, c) mfma_v(a, b, c)

template <int MODE>
__global__ __launch_bounds__(256, 1) void chain(const bf16x8* __restrict__ in, f32x4* __restrict__ out, long long* __restrict__ cycles, int iters) {
    const int t = threadIdx.x;
    bf16x8 wh[4], wl[4], ah0 = in[t], al0 = in[256 + t], ah1 = in[512 + t], al1 = in[768 + t];
#pragma unroll
    for (int i = 0; i < 4; ++i) { wh[i] = in[1024 + 256 * i + t]; wl[i] = in[2048 + 256 * i + t]; }
    f32x4 c[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; c[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float f[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) f[i] = 1.0f + 1e-3f * (float)(t + i);
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    u32x4_t ld[4] = {};
    const unsigned lds_addr = (unsigned)(size_t)(lds) + (unsigned)t * 16u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(in), 0, 4u << 20, 0x00020000);
    if (t < 16) reinterpret_cast<float*>(lds)[t] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {            // the product kernel's order: one weight tile at a time, accumulators alternate (distance 2)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[i][0] = MFMA(wl[i & 3], ah0, c[i][0]); c[i][1] = MFMA(wl[i & 3], ah1, c[i][1]);
                c[i][0] = MFMA(wh[i & 3], al0, c[i][0]); c[i][1] = MFMA(wh[i & 3], al1, c[i][1]);
                c[i][0] = MFMA(wh[i & 3], ah0, c[i][0]); c[i][1] = MFMA(wh[i & 3], ah1, c[i][1]);
            }
        } else if constexpr (MODE == 1) {     // two weight tiles interleaved: distance 4
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                c[i][0] = MFMA(wl[i & 3], ah0, c[i][0]); c[i][1] = MFMA(wl[i & 3], ah1, c[i][1]);
                c[i + 1][0] = MFMA(wl[(i + 1) & 3], ah0, c[i + 1][0]); c[i + 1][1] = MFMA(wl[(i + 1) & 3], ah1, c[i + 1][1]);
                c[i][0] = MFMA(wh[i & 3], al0, c[i][0]); c[i][1] = MFMA(wh[i & 3], al1, c[i][1]);
                c[i + 1][0] = MFMA(wh[(i + 1) & 3], al0, c[i + 1][0]); c[i + 1][1] = MFMA(wh[(i + 1) & 3], al1, c[i + 1][1]);
                c[i][0] = MFMA(wh[i & 3], ah0, c[i][0]); c[i][1] = MFMA(wh[i & 3], ah1, c[i][1]);
                c[i + 1][0] = MFMA(wh[(i + 1) & 3], ah0, c[i + 1][0]); c[i + 1][1] = MFMA(wh[(i + 1) & 3], ah1, c[i + 1][1]);
            }
        } else if constexpr (MODE == 2) {     // term-major over all eight tiles: distance 16
#pragma unroll
            for (int i = 0; i < 8; ++i) { c[i][0] = MFMA(wl[i & 3], ah0, c[i][0]); c[i][1] = MFMA(wl[i & 3], ah1, c[i][1]); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { c[i][0] = MFMA(wh[i & 3], al0, c[i][0]); c[i][1] = MFMA(wh[i & 3], al1, c[i][1]); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { c[i][0] = MFMA(wh[i & 3], ah0, c[i][0]); c[i][1] = MFMA(wh[i & 3], ah1, c[i][1]); }
        } else if constexpr (MODE >= 4 && MODE <= 9) {
            // the mma3_w order with filler between the MFMAs: MODE 4 / 5 / 6 = 1 / 2 / 3 independent v_fma_f32 behind each MFMA, 7 = one v_exp_f32 behind each,
            // 8 = a DEPENDENT chain of two v_fma_f32 behind each (one chain across the whole loop), 9 = two ds_read_b128 per six MFMAs (what run_pair2 issues)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const bf16x8& w = k < 2 ? wl[i & 3] : wh[i & 3];
                    const bf16x8& a = (k & 1) ? (k == 3 ? al1 : ah1) : (k == 2 ? al0 : ah0);
                    c[i][k & 1] = MFMA(w, a, c[i][k & 1]);
                    if constexpr (MODE == 4 || MODE == 5 || MODE == 6) {
#pragma unroll
                        for (int q = 0; q < MODE - 3; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(k * 3 + q) & 15]) : "v"(f[16]));
                    } else if constexpr (MODE == 7) {
                        asm volatile("v_exp_f32 %0, %0" : "+v"(f[k & 15]));
                    } else if constexpr (MODE == 8) {
                        asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(f[0]) : "v"(f[16]));
                    } else if constexpr (MODE == 9) {
                        if (k == 0 || k == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(k == 3) + 2 * (i & 1)]) : "v"(lds_addr + 4096 * ((i * 2 + (k == 3)) & 7)));
                    }
                }
                if constexpr (MODE == 9) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            }
        } else if constexpr (MODE == 10 || MODE == 11) {
            // the mma3_w order with the wave's LDS-DMA pieces of a 16 KiB weight stage (four buffer_load_dwordx4 ... lds of 1 KiB each, from an L2-resident
            // buffer): MODE 10 = the four back to back once per 48 MFMAs (the product kernel), MODE 11 = one per 12 MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 10 ? i == 0 : (i & 1) == 0) {
#pragma unroll
                    for (int q = 0; q < (MODE == 10 ? 4 : 1); ++q) {
                        const unsigned piece = (unsigned)((it * 4 + (MODE == 10 ? q : (i >> 1))) & 1023);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 4096 * (t >> 6) + 1024 * (MODE == 10 ? q : (i >> 1))), 16,
                                                                 (unsigned)t * 16u, piece * 4096u, 0, 0);
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const bf16x8& w = k < 2 ? wl[i & 3] : wh[i & 3];
                    const bf16x8& a = (k & 1) ? (k == 3 ? al1 : ah1) : (k == 2 ? al0 : ah0);
                    c[i][k & 1] = MFMA(w, a, c[i][k & 1]);
                }
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {                              // MODE 3: back-to-back dependent (distance 1): the latency itself
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[i][0] = MFMA(wl[i & 3], ah0, c[i][0]); c[i][0] = MFMA(wh[i & 3], al0, c[i][0]); c[i][0] = MFMA(wh[i & 3], ah0, c[i][0]);
                c[i][1] = MFMA(wl[i & 3], ah1, c[i][1]); c[i][1] = MFMA(wh[i & 3], al1, c[i][1]); c[i][1] = MFMA(wh[i & 3], ah1, c[i][1]);
            }
        }
    }
    const long long t1 = clock64();
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1];
#pragma unroll
    for (int i = 0; i < 17; ++i) s[0] += f[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[1] += __uint_as_float(ld[i][0] ^ ld[i][1] ^ ld[i][2] ^ ld[i][3]);
    out[blockIdx.x * 256 + t] = s;
    if (t == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE> static void run(const char* name, const bf16x8* in, f32x4* out, long long* cyc, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 10);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long h[4]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double mfmas = 48.0 * iters;      // per wave
    printf("%-44s blocks %4d: %7.2f s_memtime ticks per MFMA (wave 0 of block 0), %7.3f ms, %6.1f TFLOP/s issued\n", name, blocks,
           (double)h[0] / mfmas, ms, mfmas * 4 * blocks * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
    bf16x8* in; f32x4* out; long long* cyc;
    (void)hipMalloc(&in, 8u << 20); (void)hipMemset(in, 0, 8u << 20);
    (void)hipMalloc(&out, 1024 * 256 * sizeof(f32x4)); (void)hipMalloc(&cyc, 1024 * sizeof(long long));
    for (int blocks : {256, 512}) {
        run<0>("mma3_w order (accumulator every 2nd MFMA)", in, out, cyc, blocks);
        run<1>("two tiles interleaved (every 4th)", in, out, cyc, blocks);
        run<2>("term-major over eight tiles (every 16th)", in, out, cyc, blocks);
        run<3>("dependent back to back (every MFMA)", in, out, cyc, blocks);
        if (blocks == 256) {
            run<4>("mma3_w order + 1 independent v_fma per MFMA", in, out, cyc, blocks);
            run<5>("mma3_w order + 2 independent v_fma per MFMA", in, out, cyc, blocks);
            run<6>("mma3_w order + 3 independent v_fma per MFMA", in, out, cyc, blocks);
            run<7>("mma3_w order + 1 v_exp_f32 per MFMA", in, out, cyc, blocks);
            run<8>("mma3_w order + 2 DEPENDENT v_fma per MFMA", in, out, cyc, blocks);
            run<9>("mma3_w order + 2 ds_read_b128 per 6 MFMAs", in, out, cyc, blocks);
            run<10>("mma3_w order + 4 LDS-DMA pieces per 48 MFMAs", in, out, cyc, blocks);
            run<11>("mma3_w order + 1 LDS-DMA piece per 12 MFMAs", in, out, cyc, blocks);
        }
    }
    return 0;
}
