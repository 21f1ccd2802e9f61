// The one-launch encoder of encoder_blocks.h in the exact-tolerance arithmetic (precision bf16x3): the same persistent structure —
// one workgroup = one image = 128 rows, four waves of 32 rows at 512 registers, the fp32 residual stream resident in the
// accumulators of the proj / fc2 GEMMs from the first load to the K / V rows of the decoder — with every matrix product evaluated
// on bf16 PAIRS: an f32 operand v is carried as hi = bf16(v), lo = bf16(v - hi) and a product is three MFMAs
// lo*hi + hi*lo + hi*hi accumulated in fp32 (the dropped lo*lo term is <= 2^-16 relative).  LayerNorm, soft-max, GELU (the
// 1.2e-7 erf form, not the polynomial of the bf16 mode), biases and the residual stream are fp32.
//
// What changes against the bf16 kernel:
//   * weights come from the plan's block-planar hi | lo pack (parseq_hip.hip split_pack_kernel: 32 elements -> 64 B hi | 64 B lo,
//     byte offsets equal those of the f32 master), so a 128-byte LDS row of a stage is ONE 32-wide k-block — hi fragment at the
//     lane's offset, lo fragment at offset ^ 64 — instead of two k-steps; a GEMM slice that was one TRIPLE of 16 KiB stages is six
//     stages here.  They run as PAIRS of stages under one workgroup barrier (2 x 48 = 96 MFMAs per wave per barrier, what a
//     bf16 triple has);
//   * LDS: the K and V^T images exist twice (hi and lo planes, 70 KiB), which leaves the attention phase a ring of two pair groups
//     (64 KiB) — the prefetch distance in MFMA time is that of the bf16 kernel's two triple groups; the MLP phase runs four pair
//     groups (128 KiB), three pairs ahead;
//   * registers: the LayerNorm'd operand is 192 registers (hi + lo) next to the 192 of x, so the working set has 128 left: weight
//     fragments are read two positions ahead through three rotating (hi, lo) buffers instead of 8 + 8, the soft-max and P V run one
//     16-query row tile at a time.
// Row orders (pair permutation), the swizzled stage layout, the K / V^T image layouts and the accumulator <-> fragment identity
// are exactly encoder_blocks.h's; this file reuses its helpers.
#pragma once
#include "encoder_blocks.h"

namespace pq {
namespace x3 {

#ifndef X3_MLP_RING
#define X3_MLP_RING 4          // pair groups of the MLP phase
#endif
#ifndef X3_AHEAD
#define X3_AHEAD 2             // weight-fragment positions read ahead of the MFMAs
#endif
constexpr int STAGE = 16384, PAIRB = 2 * STAGE;
constexpr int KIMG_B = 128 * AF_KROWB, VIMG_B = 64 * AF_VROWB;           // one plane of the K / V^T image
constexpr int ATT_RING_B = 2 * PAIRB, MLP_RING_B = 4 * PAIRB;
constexpr int IMG_OFF = ATT_RING_B;                                      // K hi | K lo | V^T hi | V^T lo
constexpr int PARAM_OFF = IMG_OFF + 2 * KIMG_B + 2 * VIMG_B;             // 137216 >= MLP_RING_B
static_assert(PARAM_OFF >= MLP_RING_B, "the MLP ring must end below the parameter block");
template <int E> constexpr size_t enc_blocks_x3_lds() { return (size_t)PARAM_OFF + (size_t)(7 * E) * sizeof(float); }

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = static_cast<bf16_t>(v[i]);                                       // round to nearest even
        lo[i] = static_cast<bf16_t>(v[i] - static_cast<float>(hi[i]));           // exact residual, rounded once
    }
}

// acc[j] += A B^T for the two row tiles j of the wave, A / B given as (hi, lo) fragments; small terms first, the two row tiles
// interleaved so that no MFMA waits for the one before it
#define PQ_X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
// weights as the first operand (q, k, fc1, proj, fc2, K|V chunks)
__device__ __forceinline__ void mma3_w(f32x4& c0, f32x4& c1, const bf16x8& wh, const bf16x8& wl, const bf16x8& ah0, const bf16x8& al0,
                                       const bf16x8& ah1, const bf16x8& al1) {
    c0 = PQ_X3_MFMA(wl, ah0, c0); c1 = PQ_X3_MFMA(wl, ah1, c1);
    c0 = PQ_X3_MFMA(wh, al0, c0); c1 = PQ_X3_MFMA(wh, al1, c1);
    c0 = PQ_X3_MFMA(wh, ah0, c0); c1 = PQ_X3_MFMA(wh, ah1, c1);
}
// weights as the second operand (the v chunk: V^T)
__device__ __forceinline__ void mma3_a(f32x4& c0, f32x4& c1, const bf16x8& wh, const bf16x8& wl, const bf16x8& ah0, const bf16x8& al0,
                                       const bf16x8& ah1, const bf16x8& al1) {
    c0 = PQ_X3_MFMA(al0, wh, c0); c1 = PQ_X3_MFMA(al1, wh, c1);
    c0 = PQ_X3_MFMA(ah0, wl, c0); c1 = PQ_X3_MFMA(ah1, wl, c1);
    c0 = PQ_X3_MFMA(ah0, wh, c0); c1 = PQ_X3_MFMA(ah1, wh, c1);
}
__device__ __forceinline__ void mma3_1(f32x4& c, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
    c = PQ_X3_MFMA(al, bh, c); c = PQ_X3_MFMA(ah, bl, c); c = PQ_X3_MFMA(ah, bh, c);
}

// Per-lane DMA source offsets in BYTES of the block-planar pack (StreamLane of encoder_blocks.h with 4-byte elements: the row
// order and the source swizzle are the same, a stage row is 128 bytes = one k-block's hi | lo halves).
struct StreamLaneX {
    unsigned v64, v128, v128w;           // 64 rows x two k-blocks at pitch 4E bytes | 128 rows x one k-block at pitch 4E | at pitch 16E
    __device__ __forceinline__ StreamLaneX(int lane, int wid, int E) {
        const int sc = ((lane & 7) ^ (lane >> 3)) * 16;
        const int rho = wid * 32 + (lane >> 3);
        const int i = rho >> 4, r16 = rho & 15, i4 = i & 3;
        const int p64 = ((i4 >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i4 & 1) * 4 + (r16 & 3);
        const int p128 = (i >> 2) * 64 + ((i >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3);
        v64 = (unsigned)(p64 * 4 * E + (rho >> 6) * 128 + sc);
        v128 = (unsigned)(p128 * 4 * E + sc);
        v128w = (unsigned)(p128 * 16 * E + sc);
    }
};
// one stage (the wave's four 1-KiB pieces): origin_b = byte offset of (row 0, k-block 0) of the stage in the pack, pitch_b = row pitch in bytes
__device__ __forceinline__ void issue_stage(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned origin_b, unsigned pitch_b, unsigned char* dst) {
    StreamLane::issue_v(rsrc, voff, origin_b, (int)(pitch_b >> 1), dst, -1);
}

// ---- a PAIR of stages under one barrier ---------------------------------------------------------------------------------------
// mma(s, i, wh, wl): the six MFMAs that consume the (hi, lo) weight fragments of tile i (16 LDS rows) of stage s.
// issue(s): the wave's LDS-DMA pieces due at stage s (one stage of a later pair).  Fragment reads run two positions ahead of the
// MFMAs through three rotating register pairs.
template <int AHEAD = 2, class Mma, class Issue>
__device__ __forceinline__ void run_pair(const unsigned char* grp, Mma&& mma, Issue&& issue) {
    const int ln = opaque_lane();
    const int fo0 = stage_frag_off(ln), fo1 = fo0 ^ 64;
    constexpr int NB = AHEAD + 1;
    bf16x8 wh[NB], wl[NB];
    static_for<0, AHEAD>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        wh[n] = *reinterpret_cast<const bf16x8*>(grp + n * 2048 + fo0); wl[n] = *reinterpret_cast<const bf16x8*>(grp + n * 2048 + fo1);
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 16>([&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n >> 3, i = n & 7, nn = n + AHEAD;
        if constexpr (i == 0) {
            issue(s);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (nn < 16) {
            const unsigned char* src = grp + (nn >> 3) * STAGE + (nn & 7) * 2048;
            wh[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo0);
            wl[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo1);
        }
        mma(s, i, wh[n % NB], wl[n % NB]);
        if constexpr (nn < 16) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

__device__ __forceinline__ void pair_fence() {          // the pair about to run has landed (caller waited vmcnt); all waves are past the previous one
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One accumulator register -> a VALU register, as an instruction the optimiser cannot merge with another read of the same value:
// the three passes of the LayerNorm below would otherwise share ONE copy of every accumulator (192 VALU registers live at once next
// to the 192 fragment registers being produced), which the allocator answers by spilling the fragments at birth.
__device__ __forceinline__ float acc_read(const float& a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
__device__ __forceinline__ void acc_read8(const f32x4& a, const f32x4& b, float (&x)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = acc_read(a[r]); x[4 + r] = acc_read(b[r]); }
}

// LayerNorm of the rows held in the accumulators -> (hi, lo) operand fragments (fp32 arithmetic, two-pass variance).
// al1_lds != nullptr: the lo fragments of row tile 1 go to the wave's LDS region ([k-block][lane] x 16 bytes) instead of al[1][.]
template <int E>
__device__ __forceinline__ void ln_acc_to_frag(const f32x4 (&acc)[E / 16][2], const float* sgam, const float* sbet, float eps, int g,
                                               bf16x8 (&ah)[2][E / 32], bf16x8 (&al)[2][E / 32], unsigned char* al1_lds = nullptr) {
    constexpr int KSTEPS = E / 32;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float x[8];
            acc_read8(acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j], x);
            s1 += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
        }
        s1 = rows4_sum(s1);
        const float mean = s1 * (1.0f / E);
        float s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float x[8];
            acc_read8(acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j], x);
            float d[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) d[r] = x[r] - mean;
            s2 += ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) + ((d[4] * d[4] + d[5] * d[5]) + (d[6] * d[6] + d[7] * d[7]));
        }
        s2 = rows4_sum(s2);
        const float rstd = 1.0f / sqrtf(s2 * (1.0f / E) + eps);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float x[8];
            acc_read8(acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j], x);
            const float4 ga = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g + 4);
            const float4 ba = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g + 4);
            float v[8];
            v[0] = (x[0] - mean) * rstd * ga.x + ba.x; v[1] = (x[1] - mean) * rstd * ga.y + ba.y;
            v[2] = (x[2] - mean) * rstd * ga.z + ba.z; v[3] = (x[3] - mean) * rstd * ga.w + ba.w;
            v[4] = (x[4] - mean) * rstd * gb.x + bb.x; v[5] = (x[5] - mean) * rstd * gb.y + bb.y;
            v[6] = (x[6] - mean) * rstd * gb.z + bb.z; v[7] = (x[7] - mean) * rstd * gb.w + bb.w;
            if (j == 1 && al1_lds != nullptr) {
                bf16x8 lo;
                split8(v, ah[j][ks], lo);
                *reinterpret_cast<bf16x8*>(al1_lds + ks * 1024) = lo;
            } else {
                split8(v, ah[j][ks], al[j][ks]);
            }
        }
    }
}

// ---- the residual stream leaves the register file for the head loop -------------------------------------------------------------
// 192 registers per lane as 48 pieces of 16 bytes, piece-major across the workgroup's 256 lanes (every wave instruction is one
// contiguous KiB).  The buffer is private to the workgroup and re-read by the lane that wrote it.
template <int E>
__device__ __forceinline__ void park_acc(const f32x4 (&acc)[E / 16][2], float* __restrict__ dst, int tid) {
#pragma unroll
    for (int i = 0; i < E / 16; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(dst + ((size_t)(2 * i + j) * 256 + tid) * 4) = acc[i][j];
}
template <int E>
__device__ __forceinline__ void unpark_acc(f32x4 (&acc)[E / 16][2], const float* __restrict__ src, int tid) {
#pragma unroll
    for (int i = 0; i < E / 16; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(src + ((size_t)(2 * i + j) * 256 + tid) * 4);
}

// ---- head loop: O = attention(qkv(a)) for the six heads, as (hi, lo) operand fragments of the proj GEMM, to `obuf` ----------------
// Per head nine pairs — q, k, v chunks (64 outputs x K = 384: six stages of 64 rows x two k-blocks each) — alternating between the
// two pair groups (pair m = 9 h + n of the phase lives in group m & 1), then S^T = K Q^T, the soft-max and O^T = V^T P^T from the
// K / V^T image planes.  O of head h: pieces ((2 h + j) * 2 + kb) * 2 + {0 hi, 1 lo} of `obuf` (piece-major like park_acc): exactly
// the k-block 2 h + kb operand of the proj GEMM for row tile j.  heads_prefetch must have been called; returns with no LDS-DMA in flight.
template <int E>
__device__ __forceinline__ void heads_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int wid, int h, int n, int s) {
    unsigned char* dst = ring + ((h + n) & 1) * PAIRB + s * STAGE + wid * 4096;
    const int u = n / 3, pp = n - 3 * u, t = 2 * pp + s;
    issue_stage(wrsrc, sl.v64, (wqkv_off + (unsigned)((u * E + h * 64) * E + t * 64)) * 4u, 4u * E, dst);
}
template <int E>
__device__ __forceinline__ void heads_prefetch(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int wid) {
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, 0, 0, 0);
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, 0, 0, 1);
}

template <int E, int AHEAD>
__device__ __forceinline__ void heads_phase(unsigned char* ring, unsigned char* img, const float* sbq, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off,
                                            float scale, const StreamLaneX& sl, int wid, int tid, const bf16x8 (&ah)[2][E / 32],
                                            const bf16x8 (&al)[2][E / 32], float* __restrict__ obuf) {
    constexpr int H = E / 64;
    static_assert(E == 384, "written for E = 384");
    unsigned char* kimg_h = img; unsigned char* kimg_l = img + KIMG_B;
    unsigned char* vimg_h = img + 2 * KIMG_B; unsigned char* vimg_l = vimg_h + VIMG_B;
    const float sc2 = scale * 1.44269504088896340736f;

    for (int h = 0; h < H; ++h) {
        f32x4 acc1[4][2];
        bf16x8 qh[2][2], ql[2][2];
        static_for<0, 9>([&](auto nc) {
            constexpr int n = decltype(nc)::value, u = n / 3, pp = n % 3;
            wait_vmcnt<0>();                                     // this pair (issued during the previous one) has landed
            pair_fence();
            if constexpr (pp == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            auto issue = [&](int s) {
                if constexpr (n < 8) heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, h, n + 1, s);
                else if (h + 1 < H) heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, h + 1, 0, s);
            };
            const unsigned char* grp = ring + ((h + n) & 1) * PAIRB;
            if constexpr (u < 2) {
                run_pair<AHEAD>(grp, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 4 * pp + 2 * s + (i >> 2);
                    mma3_w(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], al[1][kb]);
                }, issue);
            } else {
                run_pair<AHEAD>(grp, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 4 * pp + 2 * s + (i >> 2);
                    mma3_a(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], al[1][kb]);
                }, issue);
            }
            if constexpr (u < 2 && pp == 2) {
                // q -> fragments, k -> the K image planes (rows in the order the P fragments need: encoder_attn_fused.h)
                const int ln = opaque_lane();
                const int rr = ln & 15, g = ln >> 4;
                const int krow_j0 = 32 * wid + 16 * ((rr >> 2) & 1) + 4 * (rr >> 3) + (rr & 3);
                const float* bp0 = sbq + u * E + h * 64 + 8 * g;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc1[2 * pr][j][r] + bp0[32 * pr + r];
                            v[4 + r] = acc1[2 * pr + 1][j][r] + bp0[32 * pr + 4 + r];
                        }
                        bf16x8 fh, fl;
                        split8(v, fh, fl);
                        if constexpr (u == 0) { qh[j][pr] = fh; ql[j][pr] = fl; }
                        else {
                            const int off = (krow_j0 + 8 * j) * AF_KROWB + 64 * pr + 16 * g;
                            *reinterpret_cast<bf16x8*>(kimg_h + off) = fh;
                            *reinterpret_cast<bf16x8*>(kimg_l + off) = fl;
                        }
                    }
            } else if constexpr (u == 2 && pp == 2) {
                const int ln = opaque_lane();
                const int rr = ln & 15, g = ln >> 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float bv = sbq[2 * E + h * 64 + ((i >> 1) & 1) * 32 + (rr >> 2) * 8 + (i & 1) * 4 + (rr & 3)];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bf16x4 fh, fl;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc1[i][j][r] + bv;
                            fh[r] = static_cast<bf16_t>(v);
                            fl[r] = static_cast<bf16_t>(v - static_cast<float>(fh[r]));
                        }
                        const int off = (16 * i + rr) * AF_VROWB + 2 * (32 * wid + 16 * j + 4 * g);
                        *reinterpret_cast<bf16x4*>(vimg_h + off) = fh;
                        *reinterpret_cast<bf16x4*>(vimg_l + off) = fl;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // S^T = K Q^T, soft-max, O^T = V^T P^T — one 16-query row tile at a time
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 sc[8];
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt) sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int kt = 0; kt < 8; ++kt) {
                            const int off = (16 * kt + rr) * AF_KROWB + 64 * ks + 16 * g;
                            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(kimg_h + off), kl = *reinterpret_cast<const bf16x8*>(kimg_l + off);
                            mma3_1(sc[kt], kh, kl, qh[j][ks], ql[j][ks]);
                        }
                    float mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
                    mx = rows4_max(mx);
                    const float mc = mx * sc2;
                    float sum = 0.f;
                    bf16x8 ph[4], pl[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = __builtin_amdgcn_exp2f(sc[2 * ks][r] * sc2 - mc);
                            v[4 + r] = __builtin_amdgcn_exp2f(sc[2 * ks + 1][r] * sc2 - mc);
                            sum += v[r] + v[4 + r];
                        }
                        split8(v, ph[ks], pl[ks]);
                    }
                    sum = rows4_sum(sum);
                    const float inv = 1.0f / sum;
                    f32x4 ov[4];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) ov[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            const int off = (16 * dt + rr) * AF_VROWB + 64 * ks + 16 * g;
                            const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vimg_h + off), vl = *reinterpret_cast<const bf16x8*>(vimg_l + off);
                            mma3_1(ov[dt], vh, vl, ph[ks], pl[ks]);
                        }
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = ov[2 * pr][r] * inv; v[4 + r] = ov[2 * pr + 1][r] * inv; }
                        bf16x8 fh, fl;
                        split8(v, fh, fl);
                        float* o = obuf + ((size_t)((((2 * h + j) * 2 + pr) * 2) * 256) + tid) * 4;
                        *reinterpret_cast<bf16x8*>(o) = fh;
                        *reinterpret_cast<bf16x8*>(o + 256 * 4) = fl;
                    }
                }
            }
        });
    }
}

// ---- proj: acc2 += Wproj O  (bias NOT added) ------------------------------------------------------------------------------------
// A K = 384 GEMM with the operand (oh, ol) resident like the LayerNorm'd operand of fc1: 36 stages of 128 rows x one k-block, stage
// t = (k-block t / 3, row group t % 3), 18 pairs through the MLP ring (pair n in group n % RING, issued during pair n - (RING - 1)).
template <int E, int RING>
__device__ __forceinline__ void proj_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, int wid, int n, int s) {
    const int t = 2 * n + s, kb = t / 3, ng = t - 3 * kb;
    unsigned char* dst = ring + (n % RING) * PAIRB + s * STAGE + wid * 4096;
    issue_stage(wrsrc, sl.v128, (wproj_off + (unsigned)(ng * 128 * E + kb * 32)) * 4u, 4u * E, dst);
}
template <int E, int RING>
__device__ __forceinline__ void proj_prefetch(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, int wid) {
    static_for<0, RING - 1>([&](auto nc) {
        proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, wid, decltype(nc)::value, 0);
        proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, wid, decltype(nc)::value, 1);
    });
}
template <int E, int RING, int AHEAD>
__device__ __forceinline__ void proj_phase(unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, const StreamLaneX& sl, int wid,
                                           const bf16x8 (&oh)[2][E / 32], const bf16x8 (&ol)[2][E / 32], f32x4 (&acc2)[E / 16][2]) {
    constexpr int NP = 3 * (E / 32) / 2, D = RING - 1;
    static_assert(E == 384, "written for E = 384");
    static_for<0, NP>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int behind = (NP - 1 - n) < (D - 1) ? (NP - 1 - n) : (D - 1);       // pairs issued after this one and still in flight
        wait_vmcnt<8 * behind>();
        pair_fence();
        run_pair<AHEAD>(ring + (n % RING) * PAIRB, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
            const int t = 2 * n + s, kb = t / 3, ng = t % 3;
            mma3_w(acc2[ng * 8 + i][0], acc2[ng * 8 + i][1], wh, wl, oh[0][kb], ol[0][kb], oh[1][kb], ol[1][kb]);
        }, [&](int s) { if constexpr (n + D < NP) proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, wid, n + D, s); });
    });
}

// ---- MLP phase: acc2 += fc2(gelu(fc1(a) + b1))  (bias of fc2 NOT added) --------------------------------------------------------
// Per 64-wide hidden chunk six pairs: fc1 (64 rows x K = 384: six stages of 64 rows x two k-blocks) and fc2 (384 outputs x K = 64:
// six stages of 128 rows x one k-block, ordered k-block-major: stage t = (k-block t / 3, row group t % 3), so that the GELU'd hidden
// fragments of ONE k-block are live at a time — the second k-block's GELU runs at the stage boundary inside the middle pair).
// Pair n of the phase (0 .. 6 * chunks) lives in group n % RING and is issued during pair n - (RING - 1).
template <int E, int RING>
__device__ __forceinline__ void mlp_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          int wid, int c, int r, int s) {
    constexpr int F = 4 * E;
    unsigned char* dst = ring + ((6 * c + r) % RING) * PAIRB + s * STAGE + wid * 4096;
    if (r < 3) issue_stage(wrsrc, sl.v64, (w1_off + (unsigned)(c * 64 * E + (2 * r + s) * 64)) * 4u, 4u * E, dst);
    else {
        const int t = 2 * (r - 3) + s, kb = t / 3, ng = t - 3 * kb;
        issue_stage(wrsrc, sl.v128w, (w2_off + (unsigned)(ng * 128 * F + c * 64 + kb * 32)) * 4u, 4u * F, dst);
    }
}
template <int E, int RING>
__device__ __forceinline__ void mlp_prefetch(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off, int wid) {
    static_for<0, RING - 1>([&](auto rc) {
        mlp_issue<E, RING>(sl, ring, wrsrc, w1_off, w2_off, wid, 0, decltype(rc)::value, 0);
        mlp_issue<E, RING>(sl, ring, wrsrc, w1_off, w2_off, wid, 0, decltype(rc)::value, 1);
    });
}

// GELU of hidden units 32 pr + [0, 32) of the chunk for both row tiles -> (hi, lo) fragments
__device__ __forceinline__ void gelu_frag(const f32x4 (&acc1)[4][2], const float* bp, int pr, bf16x8 (&hh)[2], bf16x8 (&hl)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gelu_erf(acc1[2 * pr][j][q] + bp[32 * pr + q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[4 + q] = gelu_erf(acc1[2 * pr + 1][j][q] + bp[32 * pr + 4 + q]);
        __builtin_amdgcn_sched_barrier(0);
        split8(v, hh[j], hl[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// al1_lds != nullptr: the lo fragments of row tile 1 live in the wave's LDS region (al[1][.] is not read)
template <int E, int RING, int AHEAD>
__device__ __forceinline__ void mlp_phase(unsigned char* ring, const float* sb1, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          const StreamLaneX& sl, int wid, const bf16x8 (&ah)[2][E / 32], const bf16x8 (&al)[2][E / 32],
                                          f32x4 (&acc2)[E / 16][2], const unsigned char* al1_lds = nullptr) {
    constexpr int F = 4 * E, NCH = F / 64, D = RING - 1;
    static_assert(E == 384 && (RING == 3 || RING == 4), "written for E = 384");
    for (int c = 0; c < NCH; ++c) {
        const bool last = c + 1 == NCH;
        f32x4 acc1[4][2];
        bf16x8 hh[2], hl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc1[i][0]), "+a"(acc1[i][1]));      // the chunk accumulators belong in the accumulator half of the file
        }
        static_for<0, 6>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            // in flight behind this pair: the next D - 1 pairs (8 pieces per wave each), fewer at the end of the phase
            if (!last) wait_vmcnt<8 * (D - 1)>(); else wait_vmcnt<8 * ((5 - r) < (D - 1) ? (5 - r) : (D - 1))>();
            pair_fence();
            const int g = opaque_lane() >> 4;
            const float* bp = sb1 + c * 64 + 8 * g;
            auto issue = [&](int s) {
                if constexpr (r + D < 6) mlp_issue<E, RING>(sl, ring, wrsrc, w1_off, w2_off, wid, c, r + D, s);
                else if (!last) mlp_issue<E, RING>(sl, ring, wrsrc, w1_off, w2_off, wid, c + 1, r + D - 6, s);
                if constexpr (r == 4) { if (s == 1) gelu_frag(acc1, bp, 1, hh, hl); }      // (k-block 0, row group 2) was stage 0 of this pair
            };
            const unsigned char* grp = ring + ((6 * c + r) % RING) * PAIRB;
            if constexpr (r < 3) {
                if (al1_lds != nullptr) {
                    bf16x8 l1[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) l1[q] = *reinterpret_cast<const bf16x8*>(al1_lds + (4 * r + q) * 1024);
                    run_pair<AHEAD>(grp, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                        const int kb = 4 * r + 2 * s + (i >> 2);
                        mma3_w(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], l1[2 * s + (i >> 2)]);
                    }, issue);
                } else
                run_pair<AHEAD>(grp, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 4 * r + 2 * s + (i >> 2);
                    mma3_w(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], al[1][kb]);
                }, issue);
            } else {
                run_pair<AHEAD>(grp, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int t = 2 * (r - 3) + s, ng = t % 3;
                    mma3_w(acc2[ng * 8 + i][0], acc2[ng * 8 + i][1], wh, wl, hh[0], hl[0], hh[1], hl[1]);
                }, issue);
            }
            if constexpr (r == 2) gelu_frag(acc1, bp, 0, hh, hl);
        });
    }
}

// ---- tail: K | V = LayerNorm_final(x) Wkv^T + bkv, head-split f32 [B][heads][128][32] (the storage type of precision bf16x3) ----
template <int E>
__device__ __forceinline__ void kv_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, int wid, int m, int s) {
    // pair m = 3 c + pp of the tail, stage s
    const int c = m / 3, pp = m - 3 * c;
    unsigned char* dst = ring + (m & 1) * PAIRB + s * STAGE + wid * 4096;
    issue_stage(wrsrc, sl.v64, (wkv_off + (unsigned)(c * 64 * E + (2 * pp + s) * 64)) * 4u, 4u * E, dst);
}
template <int E>
__device__ __forceinline__ void kv_phase(unsigned char* ring, const float* sbkv, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, const StreamLaneX& sl,
                                         int wid, int image, int heads, float* __restrict__ kmem, float* __restrict__ vmem,
                                         const bf16x8 (&ah)[2][E / 32], const bf16x8 (&al)[2][E / 32]) {
    constexpr int NC = 2 * E / 64;
    for (int c = 0; c < NC; ++c) {
        f32x4 acc1[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        static_for<0, 3>([&](auto pc) {
            constexpr int pp = decltype(pc)::value;
            wait_vmcnt<0>();
            pair_fence();
            const int m = 3 * c + pp;
            run_pair(ring + (m & 1) * PAIRB, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                const int kb = 4 * pp + 2 * s + (i >> 2);
                mma3_w(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], al[1][kb]);
            }, [&](int s) { if (m + 1 < 3 * NC) kv_issue<E>(sl, ring, wrsrc, wkv_off, wid, m + 1, s); });
        });
        const int ln = opaque_lane();
        const int rr = ln & 15, g = ln >> 4;
        float* dst = c < NC / 2 ? kmem : vmem;
        const int cc = c < NC / 2 ? c : c - NC / 2;
        const float* bp0 = sbkv + c * 64 + 8 * g;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int token = 32 * wid + 16 * j + rr;
                float* o = dst + (((size_t)image * heads + 2 * cc + pr) * 128 + token) * 32 + 8 * g;
                *reinterpret_cast<float4*>(o) = make_float4(acc1[2 * pr][j][0] + bp0[32 * pr], acc1[2 * pr][j][1] + bp0[32 * pr + 1],
                                                            acc1[2 * pr][j][2] + bp0[32 * pr + 2], acc1[2 * pr][j][3] + bp0[32 * pr + 3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(acc1[2 * pr + 1][j][0] + bp0[32 * pr + 4], acc1[2 * pr + 1][j][1] + bp0[32 * pr + 5],
                                                                acc1[2 * pr + 1][j][2] + bp0[32 * pr + 6], acc1[2 * pr + 1][j][3] + bp0[32 * pr + 7]);
            }
    }
}

// Tail parameters: element offsets into the pack / the f32 master like EncBlockParams; kmem == nullptr: no tail, x is stored.
struct EncTailX3 {
    unsigned norm_w, norm_b, wkv, bkv;
    float* kmem; float* vmem;
    int heads;
};

// wpack: the block-planar hi | lo copy of the f32 master (`wbytes` = 4 bytes per master element); pbase: the f32 master (vectors)
template <int E>
__global__ __launch_bounds__(256, 1)
void enc_blocks_x3_kernel(float* __restrict__ x, const unsigned char* __restrict__ wpack, unsigned wbytes, const float* __restrict__ pbase,
                          const EncBlockParams* __restrict__ blocks, int depth, float eps, int M, float* __restrict__ scratch, const EncTailX3 tail) {
    constexpr int F = 4 * E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    unsigned char* img = smem + IMG_OFF;
    float* sp = reinterpret_cast<float*>(smem + PARAM_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const StreamLaneX sl(lane, wid, E);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wpack), 0, wbytes, 0x00020000);

    f32x4 acc[E / 16][2];
    bf16x8 ah[2][E / 32], al[2][E / 32];
    load_x_to_acc<E>(x, m0, M, wid, rr, g, acc);
    float* xbuf = scratch + (size_t)blockIdx.x * (2 * 48 * 1024);      // 48 pieces x 256 lanes x 4 floats: the parked residual stream
    float* obuf = xbuf + 48 * 1024;                                     // ... and the attention output fragments

    for (int l = 0; l < depth; ++l) {
        const EncBlockParams* bp = blocks + l;
        // ---- attention branch, head loop: parameters bqkv (3E) | bproj (E) | ln1 gamma (E) | ln1 beta (E)
        __syncthreads();
        heads_prefetch<E>(sl, ring, wrsrc, bp->wqkv, wid);
        params_to_lds(sp, pbase + bp->bqkv, 3 * E, tid);
        params_to_lds(sp + 3 * E, pbase + bp->bproj, E, tid);
        params_to_lds(sp + 4 * E, pbase + bp->ln1_w, E, tid);
        params_to_lds(sp + 5 * E, pbase + bp->ln1_b, E, tid);
        __syncthreads();
#ifdef X3_MARK
        asm volatile("; X3MARK LN1");
#endif
        ln_acc_to_frag<E>(acc, sp + 4 * E, sp + 5 * E, eps, g, ah, al);
#ifdef X3_MARK
        asm volatile("; X3MARK PARK");
#endif
        park_acc<E>(acc, xbuf, tid);
#ifdef X3_MARK
        asm volatile("; X3MARK HEADS");
#endif
        heads_phase<E, X3_AHEAD>(ring, img, sp, wrsrc, bp->wqkv, 0.125f, sl, wid, tid, ah, al, obuf);
        // ---- attention branch, proj: x and the O fragments come back (each lane re-reads what it wrote)
        __syncthreads();                                                // every wave is done with the K / V^T images and the ring
#ifdef X3_MARK
        asm volatile("; X3MARK UNPARK");
#endif
        proj_prefetch<E, X3_MLP_RING>(sl, ring, wrsrc, bp->wproj, wid);
        // (the addresses go through an empty asm: the optimiser must not forward the stored values to these loads — that would keep
        // the 192 + 32 registers alive across the head loop, the very thing the round trip is for)
        const float* xback = xbuf; const float* oback = obuf;
        asm volatile("" : "+s"(xback), "+s"(oback) :: "memory");
        unpark_acc<E>(acc, xback, tid);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kb = 0; kb < E / 32; ++kb) {
                const float* o = oback + ((size_t)(((2 * (kb >> 1) + j) * 2 + (kb & 1)) * 2 * 256) + tid) * 4;
                ah[j][kb] = *reinterpret_cast<const bf16x8*>(o);
                al[j][kb] = *reinterpret_cast<const bf16x8*>(o + 256 * 4);
            }
#ifdef X3_MARK
        asm volatile("; X3MARK PROJ");
#endif
        proj_phase<E, X3_MLP_RING, X3_AHEAD>(ring, wrsrc, bp->wproj, sl, wid, ah, al, acc);
        add_bias_to_acc<E>(sp + 3 * E, g, acc);
        // ---- MLP branch: parameters b1 (4E) | b2 (E) | ln2 gamma (E) | ln2 beta (E)
        __syncthreads();
#ifdef X3_MARK
        asm volatile("; X3MARK MLPPRE");
#endif
        mlp_prefetch<E, X3_MLP_RING>(sl, ring, wrsrc, bp->w1, bp->w2, wid);
        params_to_lds(sp, pbase + bp->b1, F, tid);
        params_to_lds(sp + F, pbase + bp->b2, E, tid);
        params_to_lds(sp + F + E, pbase + bp->ln2_w, E, tid);
        params_to_lds(sp + F + 2 * E, pbase + bp->ln2_b, E, tid);
        __syncthreads();
#ifdef X3_MARK
        asm volatile("; X3MARK LN2");
#endif
        ln_acc_to_frag<E>(acc, sp + F + E, sp + F + 2 * E, eps, g, ah, al);
#ifdef X3_MARK
        asm volatile("; X3MARK MLP");
#endif
        mlp_phase<E, X3_MLP_RING, X3_AHEAD>(ring, sp, wrsrc, bp->w1, bp->w2, sl, wid, ah, al, acc);
#ifdef X3_MARK
        asm volatile("; X3MARK MLPEND");
#endif
        add_bias_to_acc<E>(sp + F, g, acc);
    }
    if (tail.kmem == nullptr) {
        store_acc_to_x<E>(x, m0, M, wid, rr, g, acc);
        return;
    }
    // ---- tail: parameters bkv (2E) | final norm gamma (E) | beta (E)
    __syncthreads();
    kv_issue<E>(sl, ring, wrsrc, tail.wkv, wid, 0, 0);
    kv_issue<E>(sl, ring, wrsrc, tail.wkv, wid, 0, 1);
    params_to_lds(sp, pbase + tail.bkv, 2 * E, tid);
    params_to_lds(sp + 2 * E, pbase + tail.norm_w, E, tid);
    params_to_lds(sp + 3 * E, pbase + tail.norm_b, E, tid);
    __syncthreads();
    ln_acc_to_frag<E>(acc, sp + 2 * E, sp + 3 * E, eps, g, ah, al);
    kv_phase<E>(ring, sp, wrsrc, tail.wkv, sl, wid, blockIdx.x, tail.heads, tail.kmem, tail.vmem, ah, al);
}

template <int E>
inline hipError_t launch_enc_blocks_x3(hipStream_t s, float* x, const void* wpack, size_t wbytes, const float* pbase, const EncBlockParams* blocks,
                                       int depth, float eps, int M, float* scratch, const EncTailX3& tail = EncTailX3{0, 0, 0, 0, nullptr, nullptr, 0}) {
    constexpr size_t lds = enc_blocks_x3_lds<E>();
    if (wbytes >= ((size_t)1 << 32) || M % 128 != 0) return hipErrorInvalidValue;
    auto kern = enc_blocks_x3_kernel<E>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(M / 128), dim3(256), lds, s, x, reinterpret_cast<const unsigned char*>(wpack), (unsigned)wbytes, pbase, blocks, depth, eps, M, scratch, tail);
    return hipGetLastError();
}

}  // namespace x3
}  // namespace pq
