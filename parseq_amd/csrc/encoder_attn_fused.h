// Fused encoder attention branch (bf16 throughput mode, E = 384, 128 tokens = ONE image per workgroup):
//
//     x[m] += proj( softmax(q k^T / sqrt(64)) v ),   [q | k | v] = LayerNorm(x[m]) Wqkv^T + bqkv       (timm Block: x + attn(norm1(x)))
//
// as ONE kernel — the twin of encoder_mlp.h.  The unfused pipeline ran three kernels per layer (LN + qkv panel GEMM, attention,
// proj GEMM with the residual epilogue) whose only purpose for 703 MB of their 900 MB of HBM traffic per launch triple was to hand
// q, k, v and the attention output to each other (profiles/r01_pmc_hbm_traffic.md); here x is read once and written once
// (2 x 100 MB at batch 512) and nothing else leaves the CU.
//
// Structure (one workgroup = the 128 tokens of one image = 4 waves x 32 rows, one wave per SIMD, whole register file per wave):
//   * x lives in the proj accumulators for the whole kernel (the RESIDENT form of encoder_mlp.h: the pair-permuted row order of
//     the streamed weight tiles makes the accumulator layout of a tile pair identical to the operand-fragment layout), LayerNorm
//     statistics are taken from them, LayerNorm'd rows sit in registers as MFMA fragments for the full K = E depth.
//   * Heads are walked one at a time.  Per head, twelve 16 KiB weight stages stream through a 6-slot LDS ring (global_load_lds,
//     five stages in flight across raw barriers, counted vmcnt — no global stores inside the loop):
//       stages 0-2   Wq rows of the head      q[64 d x 32 rows]   = Wq . LN(x)^T      -> + bias -> bf16 B-operand fragments (registers)
//       stages 3-5   Wk rows                  k                   likewise            -> + bias -> bf16 -> K image in LDS
//       stages 6-8   Wv rows, operand roles swapped: v^T[32 rows x 64 d] = LN(x) . Wv^T -> + bias -> bf16 -> V^T image in LDS
//       (barrier)    S^T = K Q^T (16x16x32 MFMAs, keys x queries), soft-max over the 128 keys of a query held by four lanes,
//                    un-normalised probabilities packed to bf16 in place as the B operand of O^T = V^T P^T, 1 / sum folded
//                    into the bf16 conversion of O
//       stages 9-11  Wproj[:, 64 h .. 64 h + 64) x O^T accumulated onto x           (K-split of the proj GEMM over heads)
//     Every hand-over between the five GEMMs of a head is a register-to-register re-labelling: the rows of each streamed tile
//     and of the K / V^T images are stored in the pair-permuted order
//         row(16 i + r16)  <->  index 32 (i >> 1) + 8 (r16 >> 2) + 4 (i & 1) + (r16 & 3)
//     under which the four accumulator registers of the tile pair (2 p, 2 p + 1) of lane group g are exactly the eight
//     consecutive k-slots 32 p + 8 g + [0, 8) of the next MFMA's operand.
//   * Epilogue: x = accumulators + bproj, stored as whole 128-byte lines (half-row swap as in encoder_mlp.h).
//
// Rounding points are those of the unfused bf16 path (and of oracle.forward(rounding='bf16')): LayerNorm output, q, k, v, the
// un-normalised probabilities (row sum kept in f32), the attention output; everything else f32.
#pragma once
#include <type_traits>

#include "common.h"
#include "encoder_mlp.h"
#include "encoder_panel.h"

namespace pq {

constexpr int AF_BM = 128, AF_NST = 6, AF_DIST = 5, AF_STAGE_BYTES = 128 * 128;
constexpr int AF_KROWB = 64 * 2 + 16;        // K image row pitch: 64 d of bf16 + 16 -> the 16 rows of a fragment read fall on distinct 16-byte slots
constexpr int AF_VROWB = 128 * 2 + 16;       // V^T image row pitch: 128 keys of bf16 + 16

template <int E>
constexpr size_t fused_attn_lds() {
    return (size_t)AF_NST * AF_STAGE_BYTES + 128 * AF_KROWB + 64 * AF_VROWB + (size_t)(6 * E) * sizeof(float);   // ring | K | V^T | bqkv | bproj | gamma | beta
}

template <int E>
__global__ __launch_bounds__(256, 1)
void fused_attn_kernel(float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                       const bf16_t* __restrict__ Wqkv, const float* __restrict__ bqkv, const bf16_t* __restrict__ Wproj,
                       const float* __restrict__ bproj, int M, float scale) {
    constexpr int H = E / 64;                 // heads
    constexpr int KSTEPS = E / 32;            // MFMA k-steps over E
    constexpr int KS1 = E / 128;              // ring slots per 64-row projection chunk (two 64-k stages per slot)
    constexpr int NG = E / 128;               // 128-row groups of Wproj (output columns)
    constexpr int SPH = 3 * KS1 + NG;         // stages per head
    constexpr int S = H * SPH;                // total stages
    static_assert(E == 384, "written for E = 384 (three stages per projection chunk, three output row groups)");
    static_assert(SPH % AF_NST == 0, "the ring slot of a stage must not depend on the head");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                                                 // [AF_NST][128 rows][128 B], XOR-swizzled
    unsigned char* kimg = smem + AF_NST * AF_STAGE_BYTES;                       // [128 key rows][AF_KROWB]
    unsigned char* vimg = kimg + 128 * AF_KROWB;                                // [64 d rows][AF_VROWB]
    float* sbq = reinterpret_cast<float*>(vimg + 64 * AF_VROWB);                // [3E] qkv bias
    float* sbp = sbq + 3 * E;                                                   // [E] proj bias
    float* sgam = sbp + E;                                                      // [E]
    float* sbet = sgam + E;                                                     // [E]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const bool lo_half = rr < 8;
    const int m0 = blockIdx.x * AF_BM;

    // ---- weight stream -----------------------------------------------------------------------------------------------
    // stage (head h, t), always 128 LDS rows x 128 bytes, one DMA instruction = 8 LDS rows, wave w issues rows 32 w .. 32 w + 31:
    //   t < 9 : projection chunk ch = t / 3 (q, k, v), slot tt = t % 3: Wqkv rows ch E + 64 h + p64(rho & 63), k in [128 tt, +128):
    //           LDS rows [0, 64) = k-half 0, rows [64, 128) = k-half 1 of the same 64 output units
    //   t >= 9: ng = t - 9: Wproj rows 128 ng + p128(rho), k = columns [64 h, 64 h + 64)
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
    auto issue_stage = [&](int h, int t) {            // t is a compile-time constant at every call site
        unsigned char* dst = ring + (t % AF_NST) * AF_STAGE_BYTES + wid * 4096;
        if (t < 3 * KS1) {
            const int ch = t / KS1, tt = t % KS1;
            const bf16_t* base = Wqkv + (size_t)(ch * E + h * 64) * E + tt * 128 + src_chunk;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + p64[q] * E + khalf[q]),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
        } else {
            const int ng = t - 3 * KS1;
            const bf16_t* base = Wproj + (size_t)ng * 128 * E + h * 64 + src_chunk;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)p128[q] * E),
                                                 (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
        }
    };

    // ---- prologue: one memory round trip (everything requested back to back, then the weight prefetch) ---------------------
    constexpr int PVN = 6 * E;               // bqkv | bproj | gamma | beta, contiguous in LDS
    constexpr int PV = (PVN + 255) / 256;
    float pv[PV];
#pragma unroll
    for (int i = 0; i < PV; ++i) {
        const int e = min(i * 256 + tid, PVN - 1);
        pv[i] = e < 3 * E ? bqkv[e] : (e < 4 * E ? bproj[e - 3 * E] : (e < 5 * E ? gamma[e - 4 * E] : beta[e - 5 * E]));
    }
    u32x4 raw0[2][KSTEPS], raw1[2][KSTEPS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rbase = m0 + wid * 32 + j * 16 + (rr & 7);
        const float* xlo = x + (size_t)min(rbase, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
        const float* xhi = x + (size_t)min(rbase + 8, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            raw0[j][ks] = *reinterpret_cast<const u32x4*>(xlo + ks * 32);
            raw1[j][ks] = *reinterpret_cast<const u32x4*>(xhi + ks * 32);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, AF_DIST>([&](auto tc) { issue_stage(0, decltype(tc)::value); });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PV; ++i) if (i * 256 + tid < PVN) sbq[i * 256 + tid] = pv[i];
    __syncthreads();

    // ---- x -> accumulators; LayerNorm'd A fragments: lane (r16, g) of row tile j holds row 32 wid + 16 j + r16, k in [32 ks + 8 g, +8)
    bf16x8 afrag[2][KSTEPS];
    f32x4 acc2[NG * 8][2];
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
            acc2[(ks >> 2) * 8 + 2 * (ks & 3)][j] = f32x4{xa[ks].x, xa[ks].y, xa[ks].z, xa[ks].w};
            acc2[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j] = f32x4{xb[ks].x, xb[ks].y, xb[ks].z, xb[ks].w};
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
        const float rstd = __builtin_amdgcn_rsqf(s2 * (1.0f / E) + eps);
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

    // ---- main loop over heads ----------------------------------------------------------------------------------------
    const int sx = rr & 7;
    const int frag_off = rr * 128;
    const int so0 = (g ^ sx) * 16, so1 = ((4 + g) ^ sx) * 16;
    // K image row of this lane's token (32 wid + 16 j + r16): keys are stored in the pair-permuted order of the S^T tiles
    // (image row 16 kt + rho holds key 32 (kt >> 1) + 8 (rho >> 2) + 4 (kt & 1) + (rho & 3)), i.e. token 32 a + 8 b + 4 e + d lives
    // in row 32 a + 16 e + 4 b + d
    const int krow_j0 = 32 * wid + 16 * ((rr >> 2) & 1) + 4 * (rr >> 3) + (rr & 3);          // j = 0; j = 1 adds 8
    const float sc2 = scale * 1.44269504088896340736f;       // exp(scale (s - m)) = exp2(sc2 s - sc2 m)

    for (int h = 0; h < H; ++h) {
        f32x4 acc1[4][2];
        bf16x8 qfrag[2][2], ofrag[2][2];
        static_for<0, SPH>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            // stage (h, t) has landed for this wave once at most 4 x (stages issued after it) loads are outstanding: four in the
            // steady state, fewer for the last four stages of the last head
            if constexpr (t >= SPH - (AF_DIST - 1)) {
                if (h == H - 1) wait_vmcnt<4 * (SPH - 1 - t)>(); else wait_vmcnt<4 * (AF_DIST - 1)>();
            } else {
                wait_vmcnt<4 * (AF_DIST - 1)>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // stage complete in LDS for everyone; the slot of the previous stage is free
            asm volatile("" ::: "memory");
            {
                constexpr int tn = (t + AF_DIST) % SPH;                  // stage AF_DIST ahead: this head's or the next one's
                const int hn = h + (t + AF_DIST) / SPH;
                if (hn < H) issue_stage(hn, tn);
            }
            const unsigned char* st = ring + (t % AF_NST) * AF_STAGE_BYTES + frag_off;
            bf16x8 wf0[8], wf1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wf0[i] = *reinterpret_cast<const bf16x8*>(st + i * 2048 + so0);
#pragma unroll
            for (int i = 0; i < 8; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(st + i * 2048 + so1);
            if constexpr (t < 3 * KS1) {
                constexpr int ch = t / KS1, tt = t % KS1;
                if constexpr (tt == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
                // tiles 0-3: k-stage 2 tt, tiles 4-7: k-stage 2 tt + 1 of the same 64 output units
                if constexpr (ch < 2) {          // q, k: D[unit][row]
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], afrag[0][(2 * tt + (i >> 2)) * 2], acc1[i & 3][0], 0, 0, 0);
                        acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], afrag[1][(2 * tt + (i >> 2)) * 2], acc1[i & 3][1], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], afrag[0][(2 * tt + (i >> 2)) * 2 + 1], acc1[i & 3][0], 0, 0, 0);
                        acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], afrag[1][(2 * tt + (i >> 2)) * 2 + 1], acc1[i & 3][1], 0, 0, 0);
                    }
                } else {                         // v: operand roles swapped, D[row][unit] — the transposed tile the V^T image wants
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[0][(2 * tt + (i >> 2)) * 2], wf0[i], acc1[i & 3][0], 0, 0, 0);
                        acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[1][(2 * tt + (i >> 2)) * 2], wf0[i], acc1[i & 3][1], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[0][(2 * tt + (i >> 2)) * 2 + 1], wf1[i], acc1[i & 3][0], 0, 0, 0);
                        acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[1][(2 * tt + (i >> 2)) * 2 + 1], wf1[i], acc1[i & 3][1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);

                if constexpr (tt == KS1 - 1 && ch < 2) {
                    // lane (r16, g), tile pair pr: units (= d of this head) 32 pr + 8 g + [0, 8) of row r16
                    const float* bp0 = sbq + ch * E + h * 64 + 8 * g;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            bf16x8 f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                f[r] = static_cast<bf16_t>(acc1[2 * pr][j][r] + bp0[32 * pr + r]);
                                f[4 + r] = static_cast<bf16_t>(acc1[2 * pr + 1][j][r] + bp0[32 * pr + 4 + r]);
                            }
                            if constexpr (ch == 0) qfrag[j][pr] = f;
                            else *reinterpret_cast<bf16x8*>(kimg + (krow_j0 + 8 * j) * AF_KROWB + 64 * pr + 16 * g) = f;
                        }
                }
                if constexpr (tt == KS1 - 1 && ch == 2) {
                    // lane (r16, g), tile i: d = p64-order unit of LDS row 16 i + r16 (== V^T image row), tokens 32 wid + 16 j + 4 g + [0, 4)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float bv = sbq[2 * E + h * 64 + ((i >> 1) & 1) * 32 + (rr >> 2) * 8 + (i & 1) * 4 + (rr & 3)];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float o4[4] = {acc1[i][j][0] + bv, acc1[i][j][1] + bv, acc1[i][j][2] + bv, acc1[i][j][3] + bv};
                            store4<bf16_t>(reinterpret_cast<bf16_t*>(vimg + (16 * i + rr) * AF_VROWB) + 32 * wid + 16 * j + 4 * g, o4);
                        }
                    }
                    // ---- attention of this head: every wave needs all 128 keys ------------------------------------------------
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    f32x4 sc[8][2];
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt) { sc[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; sc[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int kt = 0; kt < 8; ++kt) {
                            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kimg + (16 * kt + rr) * AF_KROWB + 64 * ks + 16 * g);
                            sc[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfrag[0][ks], sc[kt][0], 0, 0, 0);
                            sc[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfrag[1][ks], sc[kt][1], 0, 0, 0);
                        }
                    // soft-max over the 128 keys of query r16 of row tile j: 32 here, the rest in lanes r16 + 16, + 32, + 48
                    bf16x8 pfrag[2][4];
                    float inv[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float mx = -INFINITY;
#pragma unroll
                        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][j][r]);
                        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                        const float mc = mx * sc2;
                        float sum = 0.f;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            bf16x8 f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float p0 = exp2f(sc[2 * ks][j][r] * sc2 - mc), p1 = exp2f(sc[2 * ks + 1][j][r] * sc2 - mc);
                                sum += p0 + p1;
                                f[r] = static_cast<bf16_t>(p0);
                                f[4 + r] = static_cast<bf16_t>(p1);
                            }
                            pfrag[j][ks] = f;             // k-slots 8 g + [0, 8) of k-step ks  ==  keys 32 ks + 8 g + [0, 8)
                        }
                        sum += __shfl_xor(sum, 16, 64);
                        sum += __shfl_xor(sum, 32, 64);
                        inv[j] = 1.0f / sum;
                    }
                    f32x4 ov[4][2];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) { ov[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vimg + (16 * dt + rr) * AF_VROWB + 64 * ks + 16 * g);
                            ov[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfrag[0][ks], ov[dt][0], 0, 0, 0);
                            ov[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfrag[1][ks], ov[dt][1], 0, 0, 0);
                        }
                    // O^T tile pair (2 p, 2 p + 1), lane group g: d = 32 p + 8 g + [0, 8) of query r16 == k-slots of proj k-step p
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            bf16x8 f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                f[r] = static_cast<bf16_t>(ov[2 * pr][j][r] * inv[j]);
                                f[4 + r] = static_cast<bf16_t>(ov[2 * pr + 1][j][r] * inv[j]);
                            }
                            ofrag[j][pr] = f;
                        }
                }
            } else {
                constexpr int ng = t - 3 * KS1;          // proj: x[:, 128 ng .. +128) += O_h . Wproj[128 ng .. +128, 64 h .. +64)^T
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc2[ng * 8 + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], ofrag[0][0], acc2[ng * 8 + i][0], 0, 0, 0);
                    acc2[ng * 8 + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], ofrag[1][0], acc2[ng * 8 + i][1], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc2[ng * 8 + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], ofrag[0][1], acc2[ng * 8 + i][0], 0, 0, 0);
                    acc2[ng * 8 + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], ofrag[1][1], acc2[ng * 8 + i][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        });
    }

    // ---- epilogue: x = accumulators + bproj (the accumulators already contain x), 8 lanes per row via the half-row swap ----
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
                pa[r] = __float_as_uint(ta[r] + sbp[cg + r]);
                pb[r] = __float_as_uint(tb[r] + sbp[cg + 4 + r]);
            }
            const u32x4 got = swap_half_rows(lo_half ? pb : pa);
            const u32x4 first = lo_half ? pa : got, second = lo_half ? got : pb;
            const int col = 32 * q32 + cbase;
            if (r_first < M) *reinterpret_cast<u32x4*>(x + (size_t)r_first * E + col) = first;
            if (r_second < M) *reinterpret_cast<u32x4*>(x + (size_t)r_second * E + col) = second;
        }
    }
}

template <int E>
inline hipError_t launch_fused_attn(hipStream_t s, float* x, const float* gamma, const float* beta, float eps, const bf16_t* Wqkv,
                                    const float* bqkv, const bf16_t* Wproj, const float* bproj, int M) {
    constexpr size_t lds = fused_attn_lds<E>();
    auto kern = fused_attn_kernel<E>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((M + AF_BM - 1) / AF_BM), dim3(256), lds, s, x, gamma, beta, eps, Wqkv, bqkv, Wproj, bproj, M,
                       0.125f);
    return hipGetLastError();
}

}  // namespace pq
