// What do TWO waves on one SIMD of an MI355X share?  Registers only (no LDS, no memory): 256 workgroups of 512 threads (eight waves, two per SIMD:
// waves w and w + 4 are SIMD partners), every wave loops over a block of NM v_mfma_f32_16x16x32_bf16 (dependent triples on one accumulator, the
// order of encoder_blocks_x3w.h mma3_w) and a block of NV VALU instructions (independent v_fma_f32, or v_exp_f32).  Variants: both waves of a SIMD
// run the same program in phase | the partner starts with its VALU block (anti-phase) | one wave only MFMAs, the partner only VALU | s_setprio.
// Prints the kernel time against the matrix pipe's own time for the MFMAs issued (16 clk each at the measured clock) and against the sum.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/two_wave tools/microbench/two_wave.hip && ./tools/microbench/two_wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ f32x4 mfma_v(const bf16x8& a, const bf16x8& b, f32x4 c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
}

// WAVES: 4 or 8 per workgroup.  ROLES: 0 every wave runs [MFMA block][VALU block]; 1 waves 0-3 only the MFMA block, waves 4-7 only the VALU block.
// ANTI: 1 = waves 4-7 run [VALU block][MFMA block].  PRIO: 1 = waves 0-3 at s_setprio 1; 2 = waves 4-7.  VK: 0 v_fma_f32 (independent, 16 registers), 1 v_exp_f32,
// 2 = a DEPENDENT v_fma chain (one register).  DEP: 1 = dependent MFMA triples (one accumulator three times), 0 = independent (accumulators round-robin).
template <int WAVES, int NM, int NV, int ROLES, int ANTI, int PRIO, int VK, int DEP>
__global__ __launch_bounds__(WAVES * 64, 1) void two(const bf16x8* __restrict__ in, f32x4* __restrict__ out, long long* __restrict__ cycles, int iters) {
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6);
    bf16x8 wh = in[t & 255], wl = in[256 + (t & 255)], ah = in[512 + (t & 255)], al = in[768 + (t & 255)];
    f32x4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float f[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) f[i] = 1.0f + 1e-3f * (float)(t + i);
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t pk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pk[i] = f32x2_t{f[i], f[i + 8]};
    const bool second = w >= 4;
    if (PRIO == 1 && !second) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && second) __builtin_amdgcn_s_setprio(1);
    auto mblock = [&]() {
#pragma unroll
        for (int i = 0; i < NM / 3; ++i) {
            if constexpr (DEP) {
                c[i & 7] = mfma_v(wl, ah, c[i & 7]); c[i & 7] = mfma_v(wh, al, c[i & 7]); c[i & 7] = mfma_v(wh, ah, c[i & 7]);
            } else {
                c[(3 * i) & 7] = mfma_v(wl, ah, c[(3 * i) & 7]); c[(3 * i + 1) & 7] = mfma_v(wh, al, c[(3 * i + 1) & 7]); c[(3 * i + 2) & 7] = mfma_v(wh, ah, c[(3 * i + 2) & 7]);
            }
        }
    };
    auto vblock = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i & 15]) : "v"(f[16]));
            else if constexpr (VK == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i & 15]));
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[0]) : "v"(f[16]));
        }
    };
    __syncthreads();
    const long long t0 = clock64();
    if (ROLES == 2) {          // interleaved: NV / NM VALU instructions behind every MFMA of the wave's own stream
        constexpr int PER = NM ? NV / NM : 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                c[(i / 3) & 7] = mfma_v(i % 3 == 0 ? wl : wh, i % 3 == 1 ? al : ah, c[(i / 3) & 7]);
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    if constexpr (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(i * PER + q) & 15]) : "v"(f[16]));
                    else if constexpr (VK == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[(i * PER + q) & 15]));
                    else if constexpr (VK == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(i * PER + q) & 3]) : "v"(f[16]));      // four dependent chains
                    else if constexpr (VK == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[(i * PER + q) & 7]) : "v"(pk[7]));      // packed fp32 (two values per lane)
                    else if constexpr (VK == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(f[(i * PER + q) & 7]) : "v"(f[8 + ((i + q) & 7)]), "v"(f[16]));
                    else asm volatile("v_rcp_f32 %0, %0" : "+v"(f[(i * PER + q) & 15]));
                }
            }
        }
    } else if (ROLES == 1) {
        if (!second) { for (int it = 0; it < iters; ++it) mblock(); }
        else { for (int it = 0; it < iters; ++it) vblock(); }
    } else if (ANTI && second) {
        for (int it = 0; it < iters; ++it) { vblock(); mblock(); }
    } else {
        for (int it = 0; it < iters; ++it) { mblock(); vblock(); }
    }
    const long long t1 = clock64();
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i];
#pragma unroll
    for (int i = 0; i < 17; ++i) s[0] += f[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[1] += pk[i][0] + pk[i][1];
    out[blockIdx.x * 512 + t] = s;
    if ((t & 63) == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
}


// The weight-fragment path of encoder_blocks_x3w.h in isolation: every wave loops over positions of [two ds_read_b128 (1 KiB each, conflict-free) for the position LA
// ahead][three dependent MFMAs consuming the position's pair]; all waves of the workgroup read the same 64 KiB of LDS.  WAVES = 4: one wave per SIMD with TWO row tiles
// (six MFMAs per pair, the four-wave kernel's ratio); WAVES = 8: one row tile (three MFMAs per pair).  BAR: 1 = a workgroup barrier every 24 positions.
template <int WAVES, int LA, int BAR>
__global__ __launch_bounds__(WAVES * 64, 1) void frag(const bf16x8* __restrict__ in, f32x4* __restrict__ out, long long* __restrict__ cycles, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    for (int i = t; i < 65536 / 16; i += WAVES * 64) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 1023];
    constexpr int RT = WAVES == 4 ? 2 : 1, NB = LA + 1, NPOS = 24;
    bf16x8 ah[RT], al[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) { ah[j] = in[512 + 64 * j + lane]; al[j] = in[768 + 64 * j + lane]; }
    f32x4 c[8][RT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) c[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned base = (unsigned)(size_t)lds + (unsigned)lane * 16u;
    bf16x8 wh[NB], wl[NB];
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int n = 0; n < LA; ++n) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wh[n]) : "v"(base), "n"((2 * n) * 1024));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wl[n]) : "v"(base), "n"((2 * n + 1) * 1024));
        }
#pragma unroll
        for (int n = 0; n < NPOS; ++n) {
            if (n + LA < NPOS) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wh[(n + LA) % NB]) : "v"(base), "n"((2 * (n + LA)) * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wl[(n + LA) % NB]) : "v"(base), "n"((2 * (n + LA) + 1) * 1024));
                // the pair of position n has landed: at most 2 LA reads (the LA positions behind it) outstanding
                if constexpr (LA == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else if constexpr (LA == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if constexpr (LA == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < RT; ++j) c[n & 7][j] = mfma_v(wl[n % NB], ah[j], c[n & 7][j]);
#pragma unroll
            for (int j = 0; j < RT; ++j) c[n & 7][j] = mfma_v(wh[n % NB], al[j], c[n & 7][j]);
#pragma unroll
            for (int j = 0; j < RT; ++j) c[n & 7][j] = mfma_v(wh[n % NB], ah[j], c[n & 7][j]);
        }
    }
    const long long t1 = clock64();
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) s += c[i][j];
    out[blockIdx.x * 512 + t] = s;
    if ((t & 63) == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
}
template <int WAVES, int LA, int BAR>
static void runf(const char* name, const bf16x8* in, f32x4* out, long long* cyc) {
    const int iters = 2000, blocks = 256;
    auto k = frag<WAVES, LA, BAR>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, cyc, 10);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double pipe = 16.0 * 72 * 2;      // per SIMD and iteration: 144 MFMAs either way
    printf("%-78s %7.3f ms | wave 0: %7.1f clk/iter | MFMA pipe %5.0f clk/iter = %5.1f %% | LDS read %5.1f B/clk/CU\n", name, ms, (double)h[0] / iters, pipe,
           100.0 * pipe / ((double)h[0] / iters), WAVES * 48.0 * 1024 / ((double)h[0] / iters));
}

// frag<> plus the weight stream: the positions read a 144 KiB ring of nine 16 KiB stages (three groups of three); every wave copies two 1-KiB pieces of each stage of the group
// TWO groups ahead (buffer_load_dwordx4 ... lds from an L2-resident 7 MiB buffer) at the stage boundaries, waits for its pieces of the group about to run (vmcnt(6)) and meets
// the other waves at a barrier per group: encoder_blocks_x3w.h mlp_phase without the GELU.  DMA: 0 no copies (barrier only), 1 as described, 2 copies but no vmcnt wait.
template <int WAVES, int LA, int DMA, int SRC = 0, int ISS = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void ringk(const bf16x8* __restrict__ in, const unsigned char* __restrict__ wts, f32x4* __restrict__ out, long long* __restrict__ cycles, int iters, unsigned span, int nomma) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    for (int i = t; i < 147456 / 16; i += WAVES * 64) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 1023];
    constexpr int RT = WAVES == 4 ? 2 : 1, NB = LA + 1, NPOS = 24, PPW = 16 / WAVES;      // pieces per wave and stage
    bf16x8 ah[RT], al[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) { ah[j] = in[512 + 64 * j + lane]; al[j] = in[768 + 64 * j + lane]; }
    f32x4 c[8][RT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) c[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wts), 0, span + (1u << 20), 0x00020000);
    const unsigned lbase = (unsigned)(size_t)lds + (unsigned)lane * 16u;
    bf16x8 wh[NB], wl[NB];
    auto issue1 = [&](int grp, int st, int it, int w) {      // wave w's pieces of stage st of ring group grp
        if (DMA == 0) return;
        unsigned char* dst = lds + grp * 49152 + st * 16384 + w * PPW * 1024;
        auto* l = (__attribute__((address_space(3))) void*)dst;
        if constexpr (SRC == 0) {
            const unsigned src = (unsigned)(((unsigned)(it * 3 + st) * 16384u) % span) + (unsigned)(w * PPW * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (unsigned)lane * 16u, src, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (unsigned)lane * 16u, src, 1024, 0);
            if constexpr (PPW == 4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (unsigned)lane * 16u, src, 2048, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (unsigned)lane * 16u, src, 3072, 0);
            }
        } else {
            // the product kernels' pattern: a piece = 8 rows x 128 B of a [rows][1536 B] matrix (lane >> 3 = row, lane & 7 = 16-byte chunk), chunks XOR-swizzled for SRC == 2
            // SRC == 3: the fc2 stages' pattern — 128 rows x 128 B of a [384 rows][6144 B] matrix, k-block (it * 3 + st) % 48
            constexpr unsigned PITCH = SRC == 3 ? 6144u : 1536u;
            const unsigned chunk = SRC >= 2 ? (unsigned)((lane & 7) ^ (lane >> 3)) : (unsigned)(lane & 7);
            const unsigned voff = (unsigned)(lane >> 3) * PITCH + chunk * 16u;
            const unsigned g3 = (unsigned)(it * 3 + st);
            const unsigned src = SRC == 3 ? (unsigned)((((g3 / 144u) * 2359296u) % span) + ((g3 / 48u) % 3u) * 128u * PITCH + (g3 % 48u) * 128u + (unsigned)(w * PPW * 8) * PITCH)
                                          : (unsigned)(((g3 * 128u) * 1536u) % span) + (unsigned)(w * PPW * 8 * 1536) + (unsigned)((it & 7) * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, src, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, src + 8 * PITCH - 1024, 1024, 0);
            if constexpr (PPW == 4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, src + 16 * PITCH - 2048, 2048, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, src + 24 * PITCH - 3072, 3072, 0);
            }
        }
    };
    auto issue = [&](int grp, int st, int it) {
        if constexpr (ISS == 1) { if (w >= 4) { issue1(grp, st, it, w); issue1(grp, st, it, w - 4); } }
        else issue1(grp, st, it, w);
    };
    __syncthreads();
    if (DMA) { for (int st = 0; st < 3; ++st) issue(0, st, 0); for (int st = 0; st < 3; ++st) issue(1, st, 1); }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int grp = it % 3;
        if (DMA == 1) {
            if (ISS == 1) { if (w >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
            else if (WAVES == 8) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned base = lbase + (unsigned)grp * 49152u;
        if (nomma) { for (int st = 0; st < 3; ++st) issue((grp + 2) % 3, st, it + 2); continue; }
#pragma unroll
        for (int n = 0; n < LA; ++n) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wh[n]) : "v"(base), "n"((2 * n) * 1024));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wl[n]) : "v"(base), "n"((2 * n + 1) * 1024));
        }
#pragma unroll
        for (int n = 0; n < NPOS; ++n) {
            if (ISS == 2 ? (n & 7) == (w & 7) : (n & 7) == 0) issue((grp + 2) % 3, n >> 3, it + 2);
            if (n + LA < NPOS) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wh[(n + LA) % NB]) : "v"(base), "n"((2 * (n + LA)) * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wl[(n + LA) % NB]) : "v"(base), "n"((2 * (n + LA) + 1) * 1024));
                if constexpr (LA == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else if constexpr (LA == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < RT; ++j) c[n & 7][j] = mfma_v(wl[n % NB], ah[j], c[n & 7][j]);
#pragma unroll
            for (int j = 0; j < RT; ++j) c[n & 7][j] = mfma_v(wh[n % NB], al[j], c[n & 7][j]);
#pragma unroll
            for (int j = 0; j < RT; ++j) c[n & 7][j] = mfma_v(wh[n % NB], ah[j], c[n & 7][j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) s += c[i][j];
    out[blockIdx.x * 512 + t] = s;
    if ((t & 63) == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
}
template <int WAVES, int LA, int DMA, int SRC = 0, int ISS = 0>
static void runr(const char* name, const bf16x8* in, const unsigned char* wts, f32x4* out, long long* cyc, unsigned span = 6u << 20, int nomma = 0) {
    const int iters = 2000, blocks = 256;
    auto k = ringk<WAVES, LA, DMA, SRC, ISS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 147456, 0, in, wts, out, cyc, 10, span, nomma);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 147456, 0, in, wts, out, cyc, iters, span, nomma);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double pipe = 16.0 * 72 * 2, clk = (double)h[0] / iters;
    printf("%-78s %7.3f ms | wave 0: %7.1f clk/group | MFMA pipe %5.0f clk/group = %5.1f %% (from the kernel time at 2.4 GHz: %5.1f %%)\n", name, ms, clk, pipe, 100.0 * pipe / clk,
           100.0 * pipe * iters / (ms * 1e-3 * 2.4e9));
}

template <int WAVES, int NM, int NV, int ROLES, int ANTI, int PRIO, int VK, int DEP = 1>
static void run(const char* name, const bf16x8* in, f32x4* out, long long* cyc) {
    const int iters = 2000, blocks = 256;
    auto k = two<WAVES, NM, NV, ROLES, ANTI, PRIO, VK, DEP>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, cyc, 10);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // per SIMD and iteration: MFMAs issued by the waves of that SIMD, VALU instructions likewise
    const int mw = ROLES == 1 ? 1 : WAVES / 4, vw = ROLES == 1 ? 1 : WAVES / 4;
    const double it_clk0 = (double)h[0] / iters, it_clk4 = WAVES == 8 ? (double)h[4] / iters : 0.0;
    const double pipe = 16.0 * NM * mw;
    printf("%-78s %7.3f ms | wave 0: %7.1f clk/iter, wave 4: %7.1f | MFMA pipe %5.0f clk/iter = %5.1f %% of wave 0's | %d MFMA + %d VALU per SIMD-iter\n", name, ms, it_clk0, it_clk4,
           pipe, 100.0 * pipe / it_clk0, NM * mw, NV * vw);
}

int main() {
    bf16x8* in; f32x4* out; long long* cyc;
    (void)hipMalloc(&in, 1u << 20); (void)hipMemset(in, 0, 1u << 20);
    (void)hipMalloc(&out, 256 * 512 * sizeof(f32x4)); (void)hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    //  WAVES NM NV ROLES ANTI PRIO VK
    run<4, 72, 0, 0, 0, 0, 0>("4 waves (1/SIMD): 72 dependent-triple MFMAs", in, out, cyc);
    run<4, 72, 0, 0, 0, 0, 0, 0>("4 waves: 72 independent MFMAs", in, out, cyc);
    run<8, 72, 0, 0, 0, 0, 0>("8 waves (2/SIMD): 72 dependent-triple MFMAs each", in, out, cyc);
    run<8, 72, 0, 0, 0, 0, 0, 0>("8 waves: 72 independent MFMAs each", in, out, cyc);
    run<8, 72, 0, 0, 0, 1, 0>("8 waves: 72 dependent-triple MFMAs each, waves 0-3 prio 1", in, out, cyc);
    run<4, 0, 200, 0, 0, 0, 0>("4 waves: 200 independent v_fma", in, out, cyc);
    run<8, 0, 200, 0, 0, 0, 0>("8 waves: 200 independent v_fma each", in, out, cyc);
    run<4, 0, 64, 0, 0, 0, 1>("4 waves: 64 v_exp", in, out, cyc);
    run<8, 0, 64, 0, 0, 0, 1>("8 waves: 64 v_exp each", in, out, cyc);
    run<8, 72, 200, 1, 0, 0, 0>("roles: waves 0-3 72 MFMAs | waves 4-7 200 v_fma", in, out, cyc);
    run<8, 72, 288, 1, 0, 0, 0>("roles: waves 0-3 72 MFMAs | waves 4-7 288 v_fma", in, out, cyc);
    run<8, 72, 288, 1, 0, 1, 0>("roles: 72 MFMAs (prio 1) | 288 v_fma", in, out, cyc);
    run<8, 72, 288, 1, 0, 2, 0>("roles: 72 MFMAs | 288 v_fma (prio 1)", in, out, cyc);
    run<8, 72, 72, 1, 0, 0, 1>("roles: waves 0-3 72 MFMAs | waves 4-7 72 v_exp", in, out, cyc);
    run<8, 72, 288, 1, 0, 0, 2>("roles: waves 0-3 72 MFMAs | waves 4-7 288 DEPENDENT v_fma", in, out, cyc);
    run<4, 72, 200, 0, 0, 0, 0>("4 waves: [72 MFMAs][200 v_fma]", in, out, cyc);
    run<8, 72, 200, 0, 0, 0, 0>("8 waves in phase: [72 MFMAs][200 v_fma] each", in, out, cyc);
    run<8, 72, 200, 0, 1, 0, 0>("8 waves anti-phase: 0-3 [M][V], 4-7 [V][M]", in, out, cyc);
    run<8, 72, 200, 0, 1, 1, 0>("8 waves anti-phase, waves 0-3 prio 1", in, out, cyc);
    run<8, 72, 200, 0, 1, 2, 0>("8 waves anti-phase, waves 4-7 prio 1", in, out, cyc);
    run<8, 72, 200, 0, 0, 1, 0>("8 waves in phase, waves 0-3 prio 1", in, out, cyc);
    run<4, 72, 216, 2, 0, 0, 0>("4 waves: 72 MFMAs with 3 v_fma behind each (interleaved)", in, out, cyc);
    run<8, 72, 72, 2, 0, 0, 0>("8 waves: 72 MFMAs each with 1 v_fma behind each", in, out, cyc);
    run<8, 72, 144, 2, 0, 0, 0>("8 waves: 72 MFMAs each with 2 v_fma behind each", in, out, cyc);
    run<8, 72, 216, 2, 0, 0, 0>("8 waves: 72 MFMAs each with 3 v_fma behind each", in, out, cyc);
    run<8, 72, 288, 2, 0, 0, 0>("8 waves: 72 MFMAs each with 4 v_fma behind each", in, out, cyc);
    run<8, 72, 216, 2, 0, 0, 2>("8 waves: 72 MFMAs each with 3 v_fma (four dependent chains) behind each", in, out, cyc);
    run<8, 72, 72, 2, 0, 0, 1>("8 waves: 72 MFMAs each with 1 v_exp behind each", in, out, cyc);
    run<8, 72, 216, 0, 0, 0, 0>("8 waves in phase: [72 MFMAs][216 v_fma] each (block form of the 3-per-MFMA case)", in, out, cyc);
    run<8, 72, 72, 2, 0, 0, 3>("8 waves: 72 MFMAs each with 1 v_pk_fma_f32 behind each", in, out, cyc);
    run<8, 72, 144, 2, 0, 0, 3>("8 waves: 72 MFMAs each with 2 v_pk_fma_f32 behind each", in, out, cyc);
    run<8, 72, 216, 2, 0, 0, 3>("8 waves: 72 MFMAs each with 3 v_pk_fma_f32 behind each", in, out, cyc);
    run<8, 72, 144, 2, 0, 0, 4>("8 waves: 72 MFMAs each with 2 v_cvt_pk_bf16_f32 behind each", in, out, cyc);
    run<8, 72, 144, 2, 0, 0, 5>("8 waves: 72 MFMAs each with 2 v_rcp_f32 behind each", in, out, cyc);
    run<8, 72, 144, 2, 0, 0, 1>("8 waves: 72 MFMAs each with 2 v_exp_f32 behind each", in, out, cyc);
    run<4, 0, 200, 0, 0, 0, 0>("(4 waves: 200 v_fma alone, again)", in, out, cyc);
    run<4, 72, 64, 0, 0, 0, 1>("4 waves: [72 MFMAs][64 v_exp]", in, out, cyc);
    run<8, 72, 64, 0, 0, 0, 1>("8 waves in phase: [72 MFMAs][64 v_exp] each", in, out, cyc);
    run<8, 72, 64, 0, 1, 0, 1>("8 waves anti-phase: [72 MFMAs][64 v_exp]", in, out, cyc);
    runf<4, 1, 0>("frag path, 4 waves x 2 row tiles, lookahead 1", in, out, cyc);
    runf<4, 2, 0>("frag path, 4 waves x 2 row tiles, lookahead 2", in, out, cyc);
    runf<8, 1, 0>("frag path, 8 waves x 1 row tile, lookahead 1", in, out, cyc);
    runf<8, 2, 0>("frag path, 8 waves x 1 row tile, lookahead 2", in, out, cyc);
    runf<8, 3, 0>("frag path, 8 waves x 1 row tile, lookahead 3", in, out, cyc);
    runf<8, 4, 0>("frag path, 8 waves x 1 row tile, lookahead 4", in, out, cyc);
    runf<8, 1, 1>("frag path, 8 waves, lookahead 1, barrier per 24 positions", in, out, cyc);
    runf<8, 2, 1>("frag path, 8 waves, lookahead 2, barrier per 24 positions", in, out, cyc);
    runf<4, 2, 1>("frag path, 4 waves, lookahead 2, barrier per 24 positions", in, out, cyc);
    unsigned char* wts; (void)hipMalloc(&wts, 100u << 20); (void)hipMemset(wts, 0, 100u << 20);
    runr<8, 1, 0>("ring, 8 waves, lookahead 1, barrier per group, no copies", in, wts, out, cyc);
    runr<8, 1, 1>("ring, 8 waves, lookahead 1, LDS-DMA two groups ahead + vmcnt wait", in, wts, out, cyc);
    runr<8, 2, 1>("ring, 8 waves, lookahead 2, LDS-DMA two groups ahead + vmcnt wait", in, wts, out, cyc);
    runr<8, 1, 2>("ring, 8 waves, lookahead 1, LDS-DMA issued, never waited for", in, wts, out, cyc);
    runr<4, 2, 0>("ring, 4 waves x 2 row tiles, lookahead 2, no copies", in, wts, out, cyc);
    runr<4, 2, 1>("ring, 4 waves x 2 row tiles, lookahead 2, LDS-DMA + vmcnt wait", in, wts, out, cyc);
    runr<8, 1, 1, 1>("ring, 8 waves, LDS-DMA sources = 8 rows x 128 B at pitch 1536", in, wts, out, cyc);
    runr<8, 1, 1, 2>("ring, 8 waves, LDS-DMA sources = 8 rows x 128 B at pitch 1536, chunks XOR-swizzled", in, wts, out, cyc);
    runr<4, 2, 1, 1>("ring, 4 waves x 2 row tiles, sources = rows at pitch 1536", in, wts, out, cyc);
    runr<4, 2, 1, 2>("ring, 4 waves x 2 row tiles, sources = rows at pitch 1536, XOR-swizzled", in, wts, out, cyc);
    runr<8, 1, 1, 0>("ring, 8 waves, contiguous sources, stream span 84 MB", in, wts, out, cyc, 84u << 20);
    runr<8, 1, 1, 2>("ring, 8 waves, swizzled row sources, stream span 84 MB", in, wts, out, cyc, 84u << 20);
    runr<8, 1, 1, 0>("skeleton (no MFMAs, no reads): 8 waves, contiguous, span 6 MB", in, wts, out, cyc, 6u << 20, 1);
    runr<8, 1, 1, 0>("skeleton: 8 waves, contiguous, span 84 MB", in, wts, out, cyc, 84u << 20, 1);
    runr<8, 1, 1, 2>("skeleton: 8 waves, swizzled rows, span 6 MB", in, wts, out, cyc, 6u << 20, 1);
    runr<8, 1, 1, 2>("skeleton: 8 waves, swizzled rows, span 84 MB", in, wts, out, cyc, 84u << 20, 1);
    runr<4, 2, 1, 2>("skeleton: 4 waves, swizzled rows, span 84 MB", in, wts, out, cyc, 84u << 20, 1);
    runr<8, 1, 1, 3>("skeleton: 8 waves, fc2 pattern (rows at pitch 6144), span 84 MB", in, wts, out, cyc, 84u << 20, 1);
    runr<8, 1, 1, 3>("ring, 8 waves, fc2 pattern (rows at pitch 6144), span 84 MB", in, wts, out, cyc, 84u << 20);
    runr<4, 2, 1, 3>("ring, 4 waves x 2 row tiles, fc2 pattern, span 84 MB", in, wts, out, cyc, 84u << 20);
    runr<8, 1, 1, 2, 0>("ring, 8 waves, swizzled rows, 84 MB: every wave issues at the stage start", in, wts, out, cyc, 84u << 20);
    runr<8, 1, 1, 2, 1>("ring, 8 waves, swizzled rows, 84 MB: only waves 4-7 issue (4 pieces per stage each)", in, wts, out, cyc, 84u << 20);
    runr<8, 1, 1, 2, 2>("ring, 8 waves, swizzled rows, 84 MB: wave w issues at position w of the stage", in, wts, out, cyc, 84u << 20);
    runr<8, 1, 1, 2, 2>("ring, 8 waves, swizzled rows, 6 MB: wave w issues at position w of the stage", in, wts, out, cyc, 6u << 20);
    return 0;
}
