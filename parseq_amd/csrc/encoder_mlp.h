// Fused encoder MLP block (bf16 throughput mode, E = 384):
//
//     x[m] += fc2( gelu( fc1( LayerNorm(x[m]) ) ) )                 x: fp32 residual stream [M, E], updated in place
//
// i.e. timm Block's `x = x + mlp(norm2(x))` as ONE kernel.  The 4E-wide hidden activation
// never leaves the CU: in the unfused pipeline it was a 201 MB write plus a 201 MB read per layer at batch 512, and its
// store drain (about 9 B/clk per CU) serialised with the MFMA phases of the fc1 kernel (profiles/r01_panel_ablation.log).
//
// Structure (one workgroup = 128 rows, 4 waves x 32 rows, ONE wave per SIMD so that each wave may use the whole
// 512-entry register file):
//   * LayerNorm'd rows live in registers as MFMA operand fragments for the whole K = E depth (as in encoder_panel.h).
//   * The hidden dimension is walked in chunks of 64 units (acc1 = 32 accumulator registers; with 128-unit chunks the
//     accumulator file is exactly full and VGPR spills go to scratch instead of to spare AGPRs).  Per chunk:
//       GEMM1  acc1[64 hidden x 32 rows] = W1[chunk] . LN(x)^T           3 ring slots of W1 (each: 64 rows x two 64-k stages)
//       GELU   + bias, packed to bf16 IN PLACE as the operand fragments of GEMM2: the pair-permuted row order in which
//              the W1 rows are laid out in LDS makes a lane's 8 outputs of a tile pair exactly the 8 consecutive k-slots
//              (hidden units) that its lane group feeds to one MFMA k-step — no LDS round trip, no cross-lane traffic.
//       GEMM2  acc2[384 out x 32 rows] += W2[:, chunk] . H^T               3 ring slots of W2 (128 out rows x 64 k each)
//   * W1 / W2 stages stream through an 8-slot LDS ring (16 KiB each) by global_load_lds, 7 stages in flight, counted vmcnt
//     across raw barriers; there are no global stores inside the loop, so the counts are loads only.
//   * Epilogue: x += acc2 + b2, read-modify-write with the 8-lanes-per-row arrangement (half-row DPP swap) so every
//     load / store instruction touches complete 128-byte lines.
#pragma once
#include <type_traits>

#include "common.h"
#include "encoder_panel.h"

namespace pq {

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int MLP_BM = 128, MLP_NST = 8, MLP_DIST = 7, MLP_STAGE_BYTES = 128 * 128, MLP_HC = 64;

// (The ablation variants of this kernel — no weight stream, no GELU, no MFMAs, no barriers, no LayerNorm prologue, phase time stamps —
// that produced profiles/r01_panel_ablation.log are in the history at commit c342b0c (the parent of the pruning commit c0ffb4f).)
// RESIDENT: the fp32 rows of x are loaded ONCE, straight into the fc2 accumulators (the pair-permuted W2 row order makes the
// accumulator layout of a tile pair identical to the LayerNorm'd operand-fragment layout: lane (r16, g) holds columns
// 32 q + 8 g + [0, 8) of row r16), LayerNorm statistics are taken from the accumulators, and the epilogue only stores:
// x crosses HBM once in each direction (203 MB per launch at M = 65 536 instead of the 340 MB measured for the
// load - LayerNorm - ... - reload - add - store form, profiles/r01_pmc_hbm_traffic.md).
template <int E, bool RESIDENT = false>
__global__ __launch_bounds__(256, 1)
void fused_mlp_kernel(float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                      const bf16_t* __restrict__ W1, const float* __restrict__ b1, const bf16_t* __restrict__ W2,
                      const float* __restrict__ b2, int M) {
    constexpr int F = 4 * E;                  // hidden width
    constexpr int KSTEPS = E / 32;            // MFMA k-steps over E
    constexpr int KS1 = E / 128;              // W1 ring slots per chunk: slot t holds k-stages 2t (LDS rows 0-63) and 2t+1 (rows 64-127)
    constexpr int NG = E / 128;               // 128-row groups of W2 (output columns)
    constexpr int KS2 = NG;                   // W2 ring slots per chunk (the chunk's 64 hidden units = one 64-k stage per row group)
    constexpr int SPC = KS1 + KS2;            // stages per chunk
    constexpr int NCH = F / MLP_HC;           // chunks
    constexpr int S = NCH * SPC;              // total stages
    static_assert(E % 128 == 0, "E must be a multiple of 128");
    static_assert(KS1 == 3 && KS2 == 3 && MLP_NST == 8, "the issue schedule below is written for six stages per chunk and eight slots");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                                                // [MLP_NST][128 rows][128 B], XOR-swizzled
    float* sb1 = reinterpret_cast<float*>(smem + MLP_NST * MLP_STAGE_BYTES);   // [F]
    float* sb2 = sb1 + F;                                                      // [E]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const bool lo_half = rr < 8;
    const int m0 = blockIdx.x * MLP_BM;

    // ---- weight stream -----------------------------------------------------------------------------------------------
    // ring slot contents for stage s = (chunk c = s / SPC, t = s % SPC), always 128 LDS rows x 128 bytes:
    //   t < KS1 :  W1 rows of the chunk (64 hidden units), TWO k-stages: LDS rows [0,64) = k in [128 t, +64),
    //              rows [64,128) = k in [128 t + 64, +64) of the SAME 64 units.  LDS row rho: unit p64(rho & 63).
    //   t >= KS1:  ng = t - KS1: W2 rows (output columns) [128 ng, +128), k = the chunk's 64 hidden units (row pitch F).
    //              LDS row rho: output column p128(rho).
    // p64 / p128 are the pair permutations that make a lane's accumulators 8 consecutive units / columns per tile pair:
    //   p128(16 i + r16) = (i>>2)*64 + ((i>>1)&1)*32 + (r16>>2)*8 + (i&1)*4 + (r16&3),   p64 = the same on 4 tiles (i < 4).
    int p128[4], p64[4], khalf[4];
    const int src_chunk = ((lane & 7) ^ (lane >> 3)) * 8;          // XOR swizzle on the source (LDS row & 7 == lane >> 3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rho = (wid * 4 + q) * 8 + (lane >> 3);
        const int i = rho >> 4, r16 = rho & 15;
        p128[q] = (i >> 2) * 64 + ((i >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3);
        const int i4 = i & 3;
        p64[q] = ((i4 >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i4 & 1) * 4 + (r16 & 3);
        khalf[q] = (rho >> 6) * 64;
    }
    auto issue_stage = [&](int c, int t, int slot) {
        unsigned char* dst = ring + slot * MLP_STAGE_BYTES + wid * 4096;
        if (t < KS1) {
            const bf16_t* base = W1 + (size_t)c * MLP_HC * E + t * 128 + src_chunk;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + p64[q] * E + khalf[q]),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
        } else {
            const int ng = t - KS1;
            const bf16_t* base = W2 + (size_t)ng * 128 * F + c * MLP_HC + src_chunk;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)p128[q] * F),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
        }
    };
    // ---- prologue: ONE memory round trip ------------------------------------------------------------------------------
    // Under load a dependent global round trip costs 2-3 us, and the first version chained eight of them (bias loop, row
    // tile 0, its gamma/beta, row tile 1, ...: 31 us per workgroup in s_memtime stamps).  Everything the prologue needs is
    // requested back to back — both row tiles of x, biases, LayerNorm affine parameters — then the weight prefetch.
    float* sgam = sb2 + E;                   // [E] LayerNorm weight
    float* sbet = sgam + E;                  // [E] LayerNorm bias
    constexpr int PVN = F + 3 * E;           // b1 | b2 | gamma | beta, contiguous in LDS
    constexpr int PV = (PVN + 255) / 256;    // floats per thread
    float pv[PV];
#pragma unroll
    for (int i = 0; i < PV; ++i) {
        const int e = min(i * 256 + tid, PVN - 1);
        pv[i] = e < F ? b1[e] : (e < F + E ? b2[e - F] : (e < F + 2 * E ? gamma[e - F - E] : beta[e - F - 2 * E]));
    }
    u32x4 raw0[2][KSTEPS], raw1[2][KSTEPS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rbase = m0 + wid * 32 + j * 16 + (rr & 7);
        const float* xlo = x + (size_t)min(rbase, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
        const float* xhi = x + (size_t)min(rbase + 8, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            raw0[j][ks] = *reinterpret_cast<const u32x4*>(xlo + ks * 32);        // a piece of row (r16 & 7)
            raw1[j][ks] = *reinterpret_cast<const u32x4*>(xhi + ks * 32);        // a piece of row (r16 & 7) + 8
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < SPC; ++s) issue_stage(0, s, s % MLP_NST);      // all of chunk 0; later chunks: see the main loop
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PV; ++i) if (i * 256 + tid < PVN) sb1[i * 256 + tid] = pv[i];   // sb1 | sb2 | sgam | sbet are contiguous
    __syncthreads();

    // ---- LayerNorm'd A fragments: lane (r16, g) of row tile j holds row 32 wid + 16 j + r16, k in [32 ks + 8 g, +8) ------
    // (coalesced 8-lanes-per-row loads + half-row swap, see encoder_panel.h)
    bf16x8 afrag[2][KSTEPS];
    f32x4 acc2[NG * 8][2];       // fc2 accumulators, 128 x E fp32 per workgroup; RESIDENT: initialised with x itself
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float4 xa[KSTEPS], xb[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const u32x4 p0 = raw0[j][ks], p1 = raw1[j][ks];
            const u32x4 got = swap_half_rows(lo_half ? p1 : p0);
            const u32x4 ev = lo_half ? p0 : got, od = lo_half ? got : p1;
            xa[ks] = make_float4(__uint_as_float(ev[0]), __uint_as_float(ev[1]), __uint_as_float(ev[2]), __uint_as_float(ev[3]));
            xb[ks] = make_float4(__uint_as_float(od[0]), __uint_as_float(od[1]), __uint_as_float(od[2]), __uint_as_float(od[3]));
            if constexpr (RESIDENT) {      // columns 32 ks + 8 g + [0, 4) / + [4, 8) of row r16 == accumulators of tile pair ks
                acc2[(ks >> 2) * 8 + 2 * (ks & 3)][j] = f32x4{xa[ks].x, xa[ks].y, xa[ks].z, xa[ks].w};
                acc2[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j] = f32x4{xb[ks].x, xb[ks].y, xb[ks].z, xb[ks].w};
            }
        }
        float s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) s1 += ((xa[ks].x + xa[ks].y) + (xa[ks].z + xa[ks].w)) + ((xb[ks].x + xb[ks].y) + (xb[ks].z + xb[ks].w));
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 * (1.0f / E);
        float s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const float d0 = xa[ks].x - mean, d1 = xa[ks].y - mean, d2 = xa[ks].z - mean, d3 = xa[ks].w - mean;
            const float d4 = xb[ks].x - mean, d5 = xb[ks].y - mean, d6 = xb[ks].z - mean, d7 = xb[ks].w - mean;
            s2 += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
        }
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = __builtin_amdgcn_rsqf(s2 * (1.0f / E) + eps);       // v_rsq_f32, 1 ulp
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const float4 a = xa[ks], b = xb[ks];
            const float4 ga = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g + 4);
            const float4 ba = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g + 4);
            bf16x8 f;
            f[0] = static_cast<bf16_t>((a.x - mean) * rstd * ga.x + ba.x); f[1] = static_cast<bf16_t>((a.y - mean) * rstd * ga.y + ba.y);
            f[2] = static_cast<bf16_t>((a.z - mean) * rstd * ga.z + ba.z); f[3] = static_cast<bf16_t>((a.w - mean) * rstd * ga.w + ba.w);
            f[4] = static_cast<bf16_t>((b.x - mean) * rstd * gb.x + bb.x); f[5] = static_cast<bf16_t>((b.y - mean) * rstd * gb.y + bb.y);
            f[6] = static_cast<bf16_t>((b.z - mean) * rstd * gb.z + bb.z); f[7] = static_cast<bf16_t>((b.w - mean) * rstd * gb.w + bb.w);
            afrag[j][ks] = f;
        }
    }

    // ---- main loop ---------------------------------------------------------------------------------------------------
    if constexpr (!RESIDENT) {
#pragma unroll
        for (int i = 0; i < NG * 8; ++i) { acc2[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const int sx = rr & 7;
    const int frag_off = rr * 128;
    const int so0 = (g ^ sx) * 16, so1 = ((4 + g) ^ sx) * 16;

    for (int c = 0; c < NCH; ++c) {
        f32x4 acc1[4][2];
        bf16x8 hfrag[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        static_for<0, SPC>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int s = c * SPC + t;
            // Issue schedule of the weight stream (per chunk c, for chunk c + 1's six stages; slot = stage index & 7):
            //   A: stages (c+1, 0..3) in the GELU gap of chunk c   (their slots held stages (c-1,4), (c-1,5), (c,0), (c,1))
            //   B: stage  (c+1, 4) at the start of stage (c, 4)    (slot of (c, 2))
            //   C: stage  (c+1, 5) at the start of stage (c, 5)    (slot of (c, 3))
            // An LDS-DMA instruction costs 100+ cycles of issue beside MFMAs and ds_reads but a fraction of that in a
            // VALU-only gap (MI355X_MICROARCH.md, per-instruction table), so two thirds of them sit behind the GELU.
            // Stage s has landed for this wave once at most 4 x (stages issued after it) loads are outstanding:
            //   t = 0: 5   1: 4   2: 3   3: 6   4: 5   5: 5     (last chunk: 5 4 3 2 1 0, nothing is issued any more)
            {
                const bool last = c == NCH - 1;
                if constexpr (t == 0) wait_vmcnt<20>();
                else if constexpr (t == 1) wait_vmcnt<16>();
                else if constexpr (t == 2) wait_vmcnt<12>();
                else if constexpr (t == 3) { if (last) wait_vmcnt<8>(); else wait_vmcnt<24>(); }
                else if constexpr (t == 4) { if (last) wait_vmcnt<4>(); else wait_vmcnt<20>(); }
                else { if (last) wait_vmcnt<0>(); else wait_vmcnt<20>(); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // stage s complete in LDS for everyone; the slot of stage s-1 is free
            asm volatile("" ::: "memory");
            if constexpr (t == 4 || t == 5) {
                if (c + 1 < NCH) issue_stage(c + 1, t, (s + SPC) & (MLP_NST - 1));
            }
            const unsigned char* st = ring + (s & (MLP_NST - 1)) * MLP_STAGE_BYTES + frag_off;
            {
            bf16x8 wf0[8], wf1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wf0[i] = *reinterpret_cast<const bf16x8*>(st + i * 2048 + so0);
#pragma unroll
            for (int i = 0; i < 8; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(st + i * 2048 + so1);
            if constexpr (t < KS1) {
                // tiles 0-3: k-stage 2t, tiles 4-7: k-stage 2t+1 of the same 64 hidden units
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], afrag[0][(2 * t + (i >> 2)) * 2], acc1[i & 3][0], 0, 0, 0);
                    acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], afrag[1][(2 * t + (i >> 2)) * 2], acc1[i & 3][1], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], afrag[0][(2 * t + (i >> 2)) * 2 + 1], acc1[i & 3][0], 0, 0, 0);
                    acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], afrag[1][(2 * t + (i >> 2)) * 2 + 1], acc1[i & 3][1], 0, 0, 0);
                }
            } else {
                constexpr int ng = t - KS1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc2[ng * 8 + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], hfrag[0][0], acc2[ng * 8 + i][0], 0, 0, 0);
                    acc2[ng * 8 + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], hfrag[1][0], acc2[ng * 8 + i][1], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc2[ng * 8 + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], hfrag[0][1], acc2[ng * 8 + i][0], 0, 0, 0);
                    acc2[ng * 8 + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], hfrag[1][1], acc2[ng * 8 + i][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }

            if constexpr (t == KS1 - 1) {
                // hidden chunk complete: + bias, exact-erf GELU, bf16 -> operand fragments of GEMM2.
                // lane (r16, g), tile pair pr: hidden units c*64 + 32 pr + 8 g + [0, 8)  ==  k-slots of k-step pr
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const float* bp = sb1 + c * MLP_HC + 32 * pr + 8 * g;
                        bf16x8 f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = static_cast<bf16_t>(gelu_poly(acc1[2 * pr][j][r] + bp[r]));
                            f[4 + r] = static_cast<bf16_t>(gelu_poly(acc1[2 * pr + 1][j][r] + bp[4 + r]));
                        }
                        hfrag[j][pr] = f;
                    }
                if (c + 1 < NCH) {
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) issue_stage(c + 1, tn, ((c + 1) * SPC + tn) & (MLP_NST - 1));
                }
            }
        });
    }

    // ---- epilogue: x += acc2 + b2  (fp32, in place; 8 lanes per row via the half-row swap) --------------------------
    // lane (r16, g), row tile j, 32-column group q32 (tile pair): columns cg = 32 q32 + 8 g + [0, 8) as piece A = [0,4), B = [4,8)
    // All old x values of a row tile are requested before the first store (the compiler cannot hoist a load above a store
    // to the same array: the naive load -> add -> store chain cost 19 us per workgroup in s_memtime stamps).
    // Row tile 1's old values are requested piece by piece while row tile 0 is being stored (each load right after the
    // piece of tile 0 whose registers it takes over): requested only after tile 0's 24 stores they would return behind all of
    // them (vector memory is in order), requested all up front they do not fit (44 spills, measured slower).
    if constexpr (RESIDENT) {
        // the accumulators already contain x: add b2, regroup to 8 lanes per row (half-row swap) and store whole 128-byte lines
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mrow = m0 + wid * 32 + j * 16;
            const int r_first = mrow + (rr & 7), r_second = r_first + 8;
            const int cbase = 8 * g + (lo_half ? 0 : 4);
#pragma unroll
            for (int q32 = 0; q32 < E / 32; ++q32) {
                const int ng = q32 >> 2, pr = q32 & 3;
                const int cg = 32 * q32 + 8 * g;
                const f32x4 ta = acc2[ng * 8 + 2 * pr][j], tb = acc2[ng * 8 + 2 * pr + 1][j];
                u32x4 pa, pb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pa[r] = __float_as_uint(ta[r] + sb2[cg + r]);
                    pb[r] = __float_as_uint(tb[r] + sb2[cg + 4 + r]);
                }
                const u32x4 got = swap_half_rows(lo_half ? pb : pa);
                const u32x4 first = lo_half ? pa : got, second = lo_half ? got : pb;
                const int col = 32 * q32 + cbase;
                if (r_first < M) *reinterpret_cast<u32x4*>(x + (size_t)r_first * E + col) = first;
                if (r_second < M) *reinterpret_cast<u32x4*>(x + (size_t)r_second * E + col) = second;
            }
        }
    } else {
    float4 old1[E / 32], old2[E / 32], nxt1[E / 32], nxt2[E / 32];
    {
        const int r_first = m0 + wid * 32 + (rr & 7);
        const int rf = min(r_first, M - 1), rs = min(r_first + 8, M - 1);
        const int cbase = 8 * g + (lo_half ? 0 : 4);
#pragma unroll
        for (int q32 = 0; q32 < E / 32; ++q32) {
            old1[q32] = *reinterpret_cast<const float4*>(x + (size_t)rf * E + 32 * q32 + cbase);
            old2[q32] = *reinterpret_cast<const float4*>(x + (size_t)rs * E + 32 * q32 + cbase);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int mrow = m0 + wid * 32 + j * 16;
        const int r_first = mrow + (rr & 7), r_second = r_first + 8;
        const int cbase = 8 * g + (lo_half ? 0 : 4);
        const int nf = min(r_first + 16, M - 1), ns = min(r_second + 16, M - 1);
#pragma unroll
        for (int q32 = 0; q32 < E / 32; ++q32) {
            const int ng = q32 >> 2, pr = q32 & 3;
            const int cg = 32 * q32 + 8 * g;
            const f32x4 ta = acc2[ng * 8 + 2 * pr][j], tb = acc2[ng * 8 + 2 * pr + 1][j];
            u32x4 pa, pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pa[r] = __float_as_uint(ta[r] + sb2[cg + r]);
                pb[r] = __float_as_uint(tb[r] + sb2[cg + 4 + r]);
            }
            const u32x4 got = swap_half_rows(lo_half ? pb : pa);
            const u32x4 first = lo_half ? pa : got, second = lo_half ? got : pb;
            const int col = 32 * q32 + cbase;
            const float4 o1 = j == 0 ? old1[q32] : nxt1[q32], o2 = j == 0 ? old2[q32] : nxt2[q32];
            if (j == 0) {
                nxt1[q32] = *reinterpret_cast<const float4*>(x + (size_t)nf * E + col);
                nxt2[q32] = *reinterpret_cast<const float4*>(x + (size_t)ns * E + col);
            }
            if (r_first < M) {
                float4 o = o1;
                o.x += __uint_as_float(first[0]); o.y += __uint_as_float(first[1]); o.z += __uint_as_float(first[2]); o.w += __uint_as_float(first[3]);
                *reinterpret_cast<float4*>(x + (size_t)r_first * E + col) = o;
            }
            if (r_second < M) {
                float4 o = o2;
                o.x += __uint_as_float(second[0]); o.y += __uint_as_float(second[1]); o.z += __uint_as_float(second[2]); o.w += __uint_as_float(second[3]);
                *reinterpret_cast<float4*>(x + (size_t)r_second * E + col) = o;
            }
        }
    }
    }
}

template <int E, bool RESIDENT = false>
inline hipError_t launch_fused_mlp(hipStream_t s, float* x, const float* gamma, const float* beta, float eps, const bf16_t* W1,
                                   const float* b1, const bf16_t* W2, const float* b2, int M) {
    const size_t lds = (size_t)MLP_NST * MLP_STAGE_BYTES + (size_t)(7 * E) * sizeof(float);     // ring | b1 | b2 | gamma | beta
    auto kern = fused_mlp_kernel<E, RESIDENT>;
    static LdsAttr attr;                // one per template instantiation
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((M + MLP_BM - 1) / MLP_BM), dim3(256), lds, s, x, gamma, beta, eps, W1, b1, W2, b2, M);
    return hipGetLastError();
}

}  // namespace pq
