// The one-launch encoder of encoder_blocks.h in the exact-tolerance arithmetic (precision bf16x3): the same persistent structure —
// one workgroup = one image = 128 rows, four waves of 32 rows at 512 registers, the fp32 residual stream resident in the
// accumulators of the proj / fc2 GEMMs from the first load to the K / V rows of the decoder — with every matrix product evaluated
// on bf16 PAIRS: an f32 operand v is carried as hi = bf16(v), lo = bf16(v - hi) and a product is three MFMAs
// lo*hi + hi*lo + hi*hi accumulated in fp32 (the dropped lo*lo term is <= 2^-16 relative).  LayerNorm, soft-max, GELU (the
// 1.2e-7 erf form, not the polynomial of the bf16 mode), biases and the residual stream are fp32.
//
// What changes against the bf16 kernel:
//   * weights come from the plan's block-planar hi | lo pack (lib_internal.h split_pack_kernel: 32 elements -> 64 B hi | 64 B lo,
//     byte offsets equal those of the f32 master), so a 128-byte LDS row of a stage is ONE 32-wide k-block — hi fragment at the
//     lane's offset, lo fragment at offset ^ 64 — instead of two k-steps; a GEMM slice that was one TRIPLE of 16 KiB stages is six
//     stages here.  They run as PAIRS of stages under one workgroup barrier (2 x 48 = 96 MFMAs per wave per barrier, what a
//     bf16 triple has);
//   * LDS: the K and V^T images exist twice (hi and lo planes, 70 KiB), which leaves the head loop a ring of five single stages
//     (80 KiB, pairs of stages per barrier); proj and the tail run pairs through three 32 KiB groups; the MLP phase (round 4) runs
//     TRIPLES of stages — 144 MFMAs per wave per barrier — through three 48 KiB groups, two triples ahead: every group of stages
//     pays ~45 % on top of its MFMA time for its wait, barrier, cold fragment reads and DMA burst, so fewer, longer groups (-2.6 %);
//   * registers: the LayerNorm'd operand is 192 registers (hi + lo) next to the 192 of x, so the working set has 128 left: weight
//     fragments are read two positions ahead through three rotating (hi, lo) buffers instead of 8 + 8, the soft-max and P V run one
//     16-query row tile at a time.
// Row orders (pair permutation), the swizzled stage layout, the K / V^T image layouts and the accumulator <-> fragment identity
// are exactly encoder_blocks.h's; this file reuses its helpers.
#pragma once
#include "encoder_blocks.h"

namespace pq {
namespace x3 {

// (The timing ablations, in-kernel phase timers and scheduling switches of this kernel lived in a lab copy removed in round 6 (`git show 186cd8e:tools/microbench/x3_lab.h`, built by tools/x3_variants.sh and timed by tools/x3_variant_bench.py of the same commit):
// profiles/r04_x3_encoder_variants.md has the results.)
#ifndef X3_MLP_RING
#define X3_MLP_RING 3          // pair groups of proj and the tail (the MLP phase itself runs triples: MLP_RING_B)
#endif
#ifndef X3_AHEAD
#define X3_AHEAD 2             // weight-fragment positions read ahead of the MFMAs
#endif
constexpr int STAGE = 16384, PAIRB = 2 * STAGE;
constexpr int KIMG_B = 128 * AF_KROWB, VIMG_B = 64 * AF_VROWB;           // one plane of the K / V^T image
// LDS map.  Head loop: a ring of HEADS_SLOTS single stages | the K hi, K lo, V^T hi, V^T lo image planes | 6E parameter floats.
// proj / MLP / tail: three 48 KiB groups (MLP: triples; proj, tail: X3_MLP_RING pair groups in the first 96 KiB) | 7E parameter floats (proj reads its bias from the head
// loop's block, which neither its ring nor the MLP's parameter block reaches).
constexpr int HEADS_SLOTS = 5;
constexpr int IMG_OFF = HEADS_SLOTS * STAGE;                             // 81920
constexpr int HEADS_PARAM_OFF = IMG_OFF + 2 * KIMG_B + 2 * VIMG_B;       // 153600
// the MLP phase runs its weight stages three to a barrier (four TRIPLES per hidden chunk: fc1 | fc1 | fc2 k-block 0 | fc2 k-block 1, the two GELU blocks exactly at group
// boundaries) through three 48 KiB ring groups; proj and the tail run PAIRS through the first 96 KiB of the same region
constexpr int MLP_RING_B = 3 * 3 * STAGE;
constexpr int MLP_PARAM_OFF = MLP_RING_B;
template <int E> constexpr size_t enc_blocks_x3_lds() {
    constexpr size_t a = (size_t)HEADS_PARAM_OFF + (size_t)(6 * E) * sizeof(float), b = (size_t)MLP_PARAM_OFF + (size_t)(7 * E) * sizeof(float);
    return a > b ? a : b;
}
// (the MLP / tail parameter block may reach into the head loop's bqkv slots — reloaded for every block — but not into the proj bias behind them, which proj reads)
static_assert(enc_blocks_x3_lds<384>() <= 163840 && MLP_RING_B + 7 * 384 * 4 <= HEADS_PARAM_OFF + 3 * 384 * 4, "LDS map");

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = static_cast<bf16_t>(v[i]);                                       // round to nearest even
        lo[i] = static_cast<bf16_t>(v[i] - static_cast<float>(hi[i]));           // exact residual, rounded once
    }
}

__device__ __forceinline__ float x3_gelu(float x) { return gelu_erf(x); }
__device__ __forceinline__ float x3_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

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
    c = PQ_X3_MFMA(al, bh, c); c = PQ_X3_MFMA(ah, bl, c);
    c = PQ_X3_MFMA(ah, bh, c);
}

// Per-lane DMA source offsets in BYTES of the block-planar pack (StreamLane of encoder_blocks.h with 4-byte elements: the row
// order and the source swizzle are the same, a stage row is 128 bytes = one k-block's hi | lo halves).
#ifndef X3_PARK_TILES
#define X3_PARK_TILES 8        // accumulator tiles (of 24) that leave the register file for the head loop
#endif
struct StreamLaneX {
    unsigned v64_, v128_, v128w_;
    __device__ __forceinline__ StreamLaneX(int lane, int wid, int E) { v64_ = calc<0>(lane, wid, E); v128_ = calc<1>(lane, wid, E); v128w_ = calc<2>(lane, wid, E); }
    // KIND 0: 64 rows x two k-blocks at pitch 4E bytes | 1: 128 rows x one k-block at pitch 4E | 2: at pitch 16E
    template <int KIND> static __device__ __forceinline__ unsigned calc(int lane, int wid, int E) {
        const int sc = ((lane & 7) ^ (lane >> 3)) * 16;
        const int rho = wid * 32 + (lane >> 3);
        const int i = rho >> 4, r16 = rho & 15, i4 = i & 3;
        if constexpr (KIND == 0) {
            const int p64 = ((i4 >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i4 & 1) * 4 + (r16 & 3);
            return (unsigned)(p64 * 4 * E + (rho >> 6) * 128 + sc);
        } else {
            const int p128 = (i >> 2) * 64 + ((i >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3);
            return (unsigned)(p128 * (KIND == 1 ? 4 : 16) * E + sc);
        }
    }
    template <int KIND> __device__ __forceinline__ unsigned voff() const { return KIND == 0 ? v64_ : (KIND == 1 ? v128_ : v128w_); }
};
// one stage (the wave's four 1-KiB pieces): origin_b = byte offset of (row 0, k-block 0) of the stage in the pack, pitch_b = row pitch in bytes
__device__ __forceinline__ void issue_stage(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned origin_b, unsigned pitch_b, unsigned char* dst, int q = -1) {
    StreamLane::issue_v(rsrc, voff, origin_b, (int)(pitch_b >> 1), dst, q);      // q = -1: all four pieces, 0..3: that piece
}

// ---- a PAIR of stages under one barrier ---------------------------------------------------------------------------------------
// mma(s, i, wh, wl): the six MFMAs that consume the (hi, lo) weight fragments of tile i (16 LDS rows) of stage s.
// issue(s): the wave's LDS-DMA pieces due at stage s (one stage of a later pair).  Fragment reads run two positions ahead of the
// MFMAs through three rotating register pairs.
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// mid(): called between the two stages (after stage 0's MFMAs)
template <int AHEAD = 2, class Mma, class Issue, class Mid = NoMid>
__device__ __forceinline__ void run_pair2(const unsigned char* st0, const unsigned char* st1, Mma&& mma, Issue&& issue, Mid&& mid = Mid{}) {
    const int ln = opaque_lane();
    const int fo0 = stage_frag_off(ln), fo1 = fo0 ^ 64;
    constexpr int NB = AHEAD + 1;
    bf16x8 wh[NB], wl[NB];
    static_for<0, AHEAD>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        wh[n] = *reinterpret_cast<const bf16x8*>(st0 + n * 2048 + fo0); wl[n] = *reinterpret_cast<const bf16x8*>(st0 + n * 2048 + fo1);
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 16>([&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n >> 3, i = n & 7, nn = n + AHEAD;
        if constexpr (n == 8) mid();
        if constexpr (i == 0) { issue(s, -1); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (nn < 16) {
            const unsigned char* src = ((nn >> 3) ? st1 : st0) + (nn & 7) * 2048;
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

template <int AHEAD = 2, class Mma, class Issue, class Mid = NoMid>
__device__ __forceinline__ void run_pair(const unsigned char* grp, Mma&& mma, Issue&& issue, Mid&& mid = Mid{}) {
    run_pair2<AHEAD>(grp, grp + STAGE, mma, issue, mid);
}

template <int N> __device__ __forceinline__ void x3_wait_vmcnt() { wait_vmcnt<N>(); }
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

// The LayerNorm arithmetic of both bf16x3 one-launch encoders, with every fused multiply-add written out (contraction off): which of a * a + b * b's two products the
// compiler fuses is its own choice and differs between the two kernels' register forms — spelled out, the four-wave and the eight-wave kernel agree bit for bit.
__device__ __forceinline__ float ln_sum8(const float (&x)[8]) {
#pragma clang fp contract(off)
    return ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
}
__device__ __forceinline__ float ln_sq8(const float (&x)[8], float mean) {
#pragma clang fp contract(off)
    float d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = x[r] - mean;
    return (__builtin_fmaf(d[0], d[0], d[1] * d[1]) + __builtin_fmaf(d[2], d[2], d[3] * d[3])) + (__builtin_fmaf(d[4], d[4], d[5] * d[5]) + __builtin_fmaf(d[6], d[6], d[7] * d[7]));
}
__device__ __forceinline__ float ln_rstd(float s2, int E, float eps) {
#pragma clang fp contract(off)
    return 1.0f / sqrtf(__builtin_fmaf(s2, 1.0f / E, eps));
}
__device__ __forceinline__ void ln_norm8(const float (&x)[8], float mean, float rstd, const float4& ga, const float4& gb, const float4& ba, const float4& bb, float (&v)[8]) {
#pragma clang fp contract(off)
    v[0] = __builtin_fmaf((x[0] - mean) * rstd, ga.x, ba.x); v[1] = __builtin_fmaf((x[1] - mean) * rstd, ga.y, ba.y);
    v[2] = __builtin_fmaf((x[2] - mean) * rstd, ga.z, ba.z); v[3] = __builtin_fmaf((x[3] - mean) * rstd, ga.w, ba.w);
    v[4] = __builtin_fmaf((x[4] - mean) * rstd, gb.x, bb.x); v[5] = __builtin_fmaf((x[5] - mean) * rstd, gb.y, bb.y);
    v[6] = __builtin_fmaf((x[6] - mean) * rstd, gb.z, bb.z); v[7] = __builtin_fmaf((x[7] - mean) * rstd, gb.w, bb.w);
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
            s1 += ln_sum8(x);
        }
        s1 = rows4_sum(s1);
        const float mean = s1 * (1.0f / E);
        float s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float x[8];
            acc_read8(acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j], x);
            s2 += ln_sq8(x, mean);
        }
        s2 = rows4_sum(s2);
        const float rstd = ln_rstd(s2, E, eps);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float x[8];
            acc_read8(acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j], x);
            const float4 ga = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g + 4);
            const float4 ba = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g + 4);
            float v[8];
            ln_norm8(x, mean, rstd, ga, gb, ba, bb, v);
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
    for (int i = 0; i < X3_PARK_TILES; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(dst + ((size_t)(2 * i + j) * 256 + tid) * 4) = acc[i][j];
}
template <int E>
__device__ __forceinline__ void unpark_acc(f32x4 (&acc)[E / 16][2], const float* __restrict__ src, int tid) {
#pragma unroll
    for (int i = 0; i < X3_PARK_TILES; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(src + ((size_t)(2 * i + j) * 256 + tid) * 4);
}

// ---- head loop: O = attention(qkv(a)) for the six heads, as (hi, lo) operand fragments of the proj GEMM, to `obuf` ----------------
// Per head nine pairs — q, k, v chunks (64 outputs x K = 384: six stages of 64 rows x two k-blocks each) — alternating between the
// two pair groups (pair m = 9 h + n of the phase lives in group m & 1), then S^T = K Q^T, the soft-max and O^T = V^T P^T from the
// K / V^T image planes.  O of head h: pieces ((2 h + j) * 2 + kb) * 2 + {0 hi, 1 lo} of `obuf` (piece-major like park_acc): exactly
// the k-block 2 h + kb operand of the proj GEMM for row tile j.  heads_prefetch must have been called; returns with no LDS-DMA in flight.
// Stage sigma = 2 (9 h + n) + s of the phase (pair n of head h, stage s) lives in ring slot sigma % HEADS_SLOTS and is issued three
// stages ahead of its pair: at the start of pair m stage 2 m + 2 may still be in flight; stages 2 m + 3 and 2 m + 4 go out at the two
// stage boundaries of pair m, into the slots the pair's opening barrier has retired.
template <int E>
__device__ __forceinline__ void heads_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int wid, int sigma, int q = -1) {
    const int m = sigma >> 1, s = sigma & 1, h = m / 9, n = m - 9 * h;
    unsigned char* dst = ring + (sigma % HEADS_SLOTS) * STAGE + wid * 4096;
    const int u = n / 3, pp = n - 3 * u, t = 2 * pp + s;
    issue_stage(wrsrc, sl.template voff<0>(), (wqkv_off + (unsigned)((u * E + h * 64) * E + t * 64)) * 4u, 4u * E, dst, q);
}
template <int E>
__device__ __forceinline__ void heads_prefetch(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int wid) {
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, 0);
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, 1);
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, 2);
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
            const int m = 9 * h + n;                             // pair index of the phase
            // this pair has landed; the stage after it may be in flight.  (vmcnt also counts the global stores issued behind that stage — parked
            // tiles, the previous head's O pieces; letting them fly too was measured and changes nothing: profiles/r04_x3_encoder_variants.md)
            if (m + 1 < 9 * H) x3_wait_vmcnt<4>(); else x3_wait_vmcnt<0>();
            pair_fence();
            if constexpr (pp == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            auto issue = [&](int s, int q) { if (2 * m + 3 + s < 18 * H) heads_issue<E>(sl, ring, wrsrc, wqkv_off, wid, 2 * m + 3 + s, q); };
            const unsigned char* st0 = ring + ((2 * m) % HEADS_SLOTS) * STAGE;
            const unsigned char* st1 = ring + ((2 * m + 1) % HEADS_SLOTS) * STAGE;
            if constexpr (u < 2) {
                run_pair2<AHEAD>(st0, st1, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 4 * pp + 2 * s + (i >> 2);
                    mma3_w(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], al[1][kb]);
                }, issue);
            } else {
                run_pair2<AHEAD>(st0, st1, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
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
                // S^T = K Q^T, soft-max, O^T = V^T P^T for both 16-query row tiles at once: every K / V^T fragment is read once
                f32x4 sc[2][8];
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) { sc[0][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; sc[1][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                {   // 16 positions (k-step ks = n / 8, key tile kt = n % 8), image fragments read X3_AHEAD positions ahead of their MFMAs
                    constexpr int NB = X3_AHEAD + 1;
                    bf16x8 fh[NB], fl[NB];
                    const int base = rr * AF_KROWB + 16 * g;
                    static_for<0, X3_AHEAD>([&](auto nc) {
                        constexpr int n = decltype(nc)::value, off = 16 * (n & 7) * AF_KROWB + 64 * (n >> 3);
                        fh[n] = *reinterpret_cast<const bf16x8*>(kimg_h + base + off); fl[n] = *reinterpret_cast<const bf16x8*>(kimg_l + base + off);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, 16>([&](auto nc) {
                        constexpr int n = decltype(nc)::value, ks = n >> 3, kt = n & 7, nn = n + X3_AHEAD;
                        if constexpr (nn < 16) {
                            constexpr int off = 16 * (nn & 7) * AF_KROWB + 64 * (nn >> 3);
                            fh[nn % NB] = *reinterpret_cast<const bf16x8*>(kimg_h + base + off); fl[nn % NB] = *reinterpret_cast<const bf16x8*>(kimg_l + base + off);
                        }
                        mma3_w(sc[0][kt], sc[1][kt], fh[n % NB], fl[n % NB], qh[0][ks], ql[0][ks], qh[1][ks], ql[1][ks]);
                        if constexpr (nn < 16) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        } else __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                bf16x8 ph[2][4], pl[2][4];
                float inv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[j][kt][r]);
                    mx = rows4_max(mx);
                    const float mc = mx * sc2;
                    float sum = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = x3_exp2(sc[j][2 * ks][r] * sc2 - mc);
                            v[4 + r] = x3_exp2(sc[j][2 * ks + 1][r] * sc2 - mc);
                            sum += v[r] + v[4 + r];
                        }
                        split8(v, ph[j][ks], pl[j][ks]);
                    }
                    sum = rows4_sum(sum);
                    inv[j] = 1.0f / sum;
                }
                f32x4 ov[2][4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { ov[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[1][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                {   // 16 positions (k-step ks = n / 4, d tile dt = n % 4)
                    constexpr int NB = X3_AHEAD + 1;
                    bf16x8 fh[NB], fl[NB];
                    const int base = rr * AF_VROWB + 16 * g;
                    static_for<0, X3_AHEAD>([&](auto nc) {
                        constexpr int n = decltype(nc)::value, off = 16 * (n & 3) * AF_VROWB + 64 * (n >> 2);
                        fh[n] = *reinterpret_cast<const bf16x8*>(vimg_h + base + off); fl[n] = *reinterpret_cast<const bf16x8*>(vimg_l + base + off);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, 16>([&](auto nc) {
                        constexpr int n = decltype(nc)::value, ks = n >> 2, dt = n & 3, nn = n + X3_AHEAD;
                        if constexpr (nn < 16) {
                            constexpr int off = 16 * (nn & 3) * AF_VROWB + 64 * (nn >> 2);
                            fh[nn % NB] = *reinterpret_cast<const bf16x8*>(vimg_h + base + off); fl[nn % NB] = *reinterpret_cast<const bf16x8*>(vimg_l + base + off);
                        }
                        mma3_w(ov[0][dt], ov[1][dt], fh[n % NB], fl[n % NB], ph[0][ks], pl[0][ks], ph[1][ks], pl[1][ks]);
                        if constexpr (nn < 16) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        } else __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = ov[j][2 * pr][r] * inv[j]; v[4 + r] = ov[j][2 * pr + 1][r] * inv[j]; }
                        bf16x8 fh, fl;
                        split8(v, fh, fl);
                        float* o = obuf + ((size_t)((((2 * h + j) * 2 + pr) * 2) * 256) + tid) * 4;
                        *reinterpret_cast<bf16x8*>(o) = fh;
                        *reinterpret_cast<bf16x8*>(o + 256 * 4) = fl;
                    }
            }
        });
    }
}

// ---- proj: acc2 += Wproj O  (bias NOT added) ------------------------------------------------------------------------------------
// A K = 384 GEMM with the operand (oh, ol) resident like the LayerNorm'd operand of fc1: 36 stages of 128 rows x one k-block, stage
// t = (k-block t / 3, row group t % 3), 18 pairs through the MLP ring (pair n in group n % RING, issued during pair n - (RING - 1)).
template <int E, int RING>
__device__ __forceinline__ void proj_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, int wid, int n, int s, int q = -1) {
    const int t = 2 * n + s, kb = t / 3, ng = t - 3 * kb;
    unsigned char* dst = ring + (n % RING) * PAIRB + s * STAGE + wid * 4096;
    issue_stage(wrsrc, sl.template voff<1>(), (wproj_off + (unsigned)(ng * 128 * E + kb * 32)) * 4u, 4u * E, dst, q);
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
        x3_wait_vmcnt<8 * behind>();
        pair_fence();
        run_pair<AHEAD>(ring + (n % RING) * PAIRB, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
            const int t = 2 * n + s, kb = t / 3, ng = t % 3;
            mma3_w(acc2[ng * 8 + i][0], acc2[ng * 8 + i][1], wh, wl, oh[0][kb], ol[0][kb], oh[1][kb], ol[1][kb]);
        }, [&](int s, int q) { if constexpr (n + D < NP) proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, wid, n + D, s, q); });
    });
}

// ---- MLP phase: acc2 += fc2(gelu(fc1(a) + b1))  (bias of fc2 NOT added): mlp_phase3 below.  Its GELU block:
// GELU of hidden units 32 pr + [0, 32) of the chunk for both row tiles -> (hi, lo) fragments
__device__ __forceinline__ void gelu_frag(const f32x4 (&acc1)[4][2], const float* bp, int pr, bf16x8 (&hh)[2], bf16x8 (&hl)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = x3_gelu(acc1[2 * pr][j][q] + bp[32 * pr + q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[4 + q] = x3_gelu(acc1[2 * pr + 1][j][q] + bp[32 * pr + 4 + q]);
        __builtin_amdgcn_sched_barrier(0);
        split8(v, hh[j], hl[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- MLP phase on TRIPLES of stages --------------------------------------------------------------------------------------------------
// Every group of stages pays a fixed price on top of its MFMAs — the vmcnt wait, the workgroup barrier and its skew, the cold fragment reads behind it (measured: ~45 % on top of a
// pair's MFMA time, profiles/r04_x3_encoder_variants.md) — so fewer, longer groups: a hidden chunk's twelve stages as four triples (144 MFMAs per wave per barrier) whose
// boundaries are exactly where the two GELU blocks sit.  NS contiguous stages from `grp`; mma(s, i, wh, wl) / issue(s, q) as in run_pair2.
template <int NS, int AHEAD = 2, class Mma, class Issue>
__device__ __forceinline__ void run_group(const unsigned char* grp, Mma&& mma, Issue&& issue) {
    const int ln = opaque_lane();
    const int fo0 = stage_frag_off(ln), fo1 = fo0 ^ 64;
    constexpr int NB = AHEAD + 1, NPOS = 8 * NS;
    bf16x8 wh[NB], wl[NB];
    static_for<0, AHEAD>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        wh[n] = *reinterpret_cast<const bf16x8*>(grp + n * 2048 + fo0); wl[n] = *reinterpret_cast<const bf16x8*>(grp + n * 2048 + fo1);
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NPOS>([&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n >> 3, i = n & 7, nn = n + AHEAD;
        if constexpr (i == 0) { issue(s, -1); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (nn < NPOS) {
            const unsigned char* src = grp + (nn >> 3) * STAGE + (nn & 7) * 2048;
            wh[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo0);
            wl[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo1);
        }
        mma(s, i, wh[n % NB], wl[n % NB]);
        if constexpr (nn < NPOS) {
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
constexpr int TRIPB = 3 * STAGE;
// triple k of chunk c (k = 0, 1: fc1 stages 3 k .. 3 k + 2; k = 2, 3: fc2 k-block k - 2, row groups 0 .. 2), stage s, into ring group (4 c + k) % 3
template <int E>
__device__ __forceinline__ void mlp3_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                           int wid, int c, int k, int s, int q = -1) {
    constexpr int F = 4 * E;
    unsigned char* dst = ring + ((4 * c + k) % 3) * TRIPB + s * STAGE + wid * 4096;
    if (k < 2) issue_stage(wrsrc, sl.template voff<0>(), (w1_off + (unsigned)(c * 64 * E + (3 * k + s) * 64)) * 4u, 4u * E, dst, q);
    else issue_stage(wrsrc, sl.template voff<2>(), (w2_off + (unsigned)(s * 128 * F + c * 64 + (k - 2) * 32)) * 4u, 4u * F, dst, q);
}
template <int E>
__device__ __forceinline__ void mlp3_prefetch(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off, int wid) {
    static_for<0, 2>([&](auto kc) {
        static_for<0, 3>([&](auto sc) { mlp3_issue<E>(sl, ring, wrsrc, w1_off, w2_off, wid, 0, decltype(kc)::value, decltype(sc)::value); });
    });
}
template <int E, int AHEAD>
__device__ __forceinline__ void mlp_phase3(unsigned char* ring, const float* sb1, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                           const StreamLaneX& sl, int wid, const bf16x8 (&ah)[2][E / 32], const bf16x8 (&al)[2][E / 32],
                                           f32x4 (&acc2)[E / 16][2]) {
    constexpr int F = 4 * E, NCH = F / 64;
    static_assert(E == 384, "written for E = 384");
    for (int c = 0; c < NCH; ++c) {
        const bool last = c + 1 == NCH;
        f32x4 acc1[4][2];
        bf16x8 hh[2], hl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc1[i][0]), "+a"(acc1[i][1]));
        }
        const int g = opaque_lane() >> 4;
        const float* bp = sb1 + c * 64 + 8 * g;
        static_for<0, 4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k >= 2) {      // hidden units 32 (k - 2) .. + 32 of the chunk — this triple's k-block — before the wait and the barrier, not behind them
                gelu_frag(acc1, bp, k - 2, hh, hl);
            }
            // in flight behind this triple: the next one (12 pieces per wave) — none behind the phase's last
            if (!last || k < 3) x3_wait_vmcnt<12>(); else x3_wait_vmcnt<0>();
            pair_fence();
            auto issue = [&](int s, int q) {      // the triple two ahead
                if constexpr (k < 2) mlp3_issue<E>(sl, ring, wrsrc, w1_off, w2_off, wid, c, k + 2, s, q);
                else if (!last) mlp3_issue<E>(sl, ring, wrsrc, w1_off, w2_off, wid, c + 1, k - 2, s, q);
            };
            const unsigned char* grp = ring + ((4 * c + k) % 3) * TRIPB;
            if constexpr (k < 2) {
                run_group<3, AHEAD>(grp, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 2 * (3 * k + s) + (i >> 2);
                    mma3_w(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], al[1][kb]);
                }, issue);
            } else {
                run_group<3, AHEAD>(grp, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    mma3_w(acc2[s * 8 + i][0], acc2[s * 8 + i][1], wh, wl, hh[0], hl[0], hh[1], hl[1]);
                }, issue);
            }
        });
    }
}

// ---- head: x = patches W_pe^T + (pos_embed + bias) in the bf16x3 arithmetic (encoder_blocks.h patch_head, three products) ---------
// 32 x 128 crops, (4, 8) patches: k-block c of a patch row is channel c's 4 x 8 pixels, and a lane's eight k-slots of it (8 g + [0, 8))
// are ONE run of eight pixels — row 4 gy + g, columns 8 gx .. 8 gx + 7.  The 384 x 96 weight is nine stages of the pack (three 128-row
// groups x three k-blocks at a 384-byte row pitch), all issued at once into the LDS the blocks have not touched yet.  The accumulators
// start from the posb table ([128][E] f32 = pos_embed + patch-embed bias, built once per plan); pixels are split into (hi, lo) like any
// other operand (u8 pixels after the reference transform (v / 255 - 0.5) / 0.5, strhub/data/module.py:78-81; bf16 pixels have lo = 0).
// The stage loads' scalar offsets go 512 bytes below the weight's origin (StreamLane::issue_v): it must start >= 128 elements into the pack.
struct EncHeadX3 {
    const void* images; int img_dtype;      // EB_IMG_F32 / EB_IMG_BF16 / EB_IMG_U8; images == nullptr: no head, x is loaded
    unsigned wpe;                           // element offset of patch_embed.proj.weight in the pack
    const float* posb;
};
constexpr unsigned X3_HEAD_MIN_WPE = 128;

template <int E>
__device__ __forceinline__ void patch_head_x3(const EncHeadX3& hp, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, int wid, int lane, int image,
                                              f32x4 (&acc)[E / 16][2]) {
    static_assert(E == 384 && 9 * STAGE <= (int)enc_blocks_x3_lds<384>(), "three 128-row groups; nine stages fit the launch's LDS");
    constexpr int PK = 96, IH = 32, IW = 128;
    const int rr = lane & 15, g = lane >> 4;
    const unsigned vpe = StreamLaneX::calc<1>(lane, wid, PK);
    static_for<0, 9>([&](auto sc) {
        constexpr int st = decltype(sc)::value, ng = st / 3, kb = st % 3;
        issue_stage(wrsrc, vpe, (hp.wpe + (unsigned)(ng * 128 * PK + kb * 32)) * 4u, 4u * PK, ring + st * STAGE + wid * 4096);
    });
    load_x_to_acc<E>(hp.posb, 0, 128, wid, rr, g, acc);
    bf16x8 ph[2][3], pl[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int token = 32 * wid + 16 * j + rr, gy = token >> 4, gx = token & 15;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t e0 = (((size_t)image * 3 + c) * IH + gy * 4 + g) * IW + gx * 8;
            float v[8];
            if (hp.img_dtype == EB_IMG_F32) {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(hp.images) + e0);
                const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(hp.images) + e0 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else if (hp.img_dtype == EB_IMG_BF16) {
                const bf16x8 f = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(hp.images) + e0);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(f[i]);
            } else {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(hp.images) + e0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned b = ((i < 4 ? u.x : u.y) >> (8 * (i & 3))) & 0xffu;
                    v[i] = ((float)b / 255.0f - 0.5f) / 0.5f;
                }
            }
            split8(v, ph[j][c], pl[j][c]);
        }
    }
    x3_wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int fo0 = stage_frag_off(opaque_lane()), fo1 = fo0 ^ 64;
    static_for<0, 9>([&](auto sc) {
        constexpr int st = decltype(sc)::value, ng = st / 3, kb = st % 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(ring + st * STAGE + i * 2048 + fo0);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(ring + st * STAGE + i * 2048 + fo1);
            mma3_w(acc[ng * 8 + i][0], acc[ng * 8 + i][1], wh, wl, ph[0][kb], pl[0][kb], ph[1][kb], pl[1][kb]);
        }
    });
}

// ---- tail: K | V = LayerNorm_final(x) Wkv^T + bkv, head-split f32 [B][heads][128][32] (the storage type of precision bf16x3) ----
template <int E, int RING>
__device__ __forceinline__ void kv_issue(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, int wid, int m, int s, int q = -1) {
    // pair m = 3 c + pp of the tail (group m % RING, issued RING - 1 pairs ahead), stage s
    const int c = m / 3, pp = m - 3 * c;
    unsigned char* dst = ring + (m % RING) * PAIRB + s * STAGE + wid * 4096;
    issue_stage(wrsrc, sl.template voff<0>(), (wkv_off + (unsigned)(c * 64 * E + (2 * pp + s) * 64)) * 4u, 4u * E, dst, q);
}
template <int E, int RING>
__device__ __forceinline__ void kv_prefetch(const StreamLaneX& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, int wid) {
    static_for<0, RING - 1>([&](auto mc) {
        kv_issue<E, RING>(sl, ring, wrsrc, wkv_off, wid, decltype(mc)::value, 0);
        kv_issue<E, RING>(sl, ring, wrsrc, wkv_off, wid, decltype(mc)::value, 1);
    });
}
template <int E, int RING, int AHEAD>
__device__ __forceinline__ void kv_phase(unsigned char* ring, const float* sbkv, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, const StreamLaneX& sl,
                                         int wid, int image, int heads, float* __restrict__ kmem, float* __restrict__ vmem, size_t plane_elems,
                                         const bf16x8 (&ah)[2][E / 32], const bf16x8 (&al)[2][E / 32]) {
    constexpr int NC = 2 * E / 64, NP = 3 * NC, D = RING - 1;
    for (int c = 0; c < NC; ++c) {
        f32x4 acc1[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        static_for<0, 3>([&](auto pc) {
            constexpr int pp = decltype(pc)::value;
            const int m = 3 * c + pp;
            static_for<0, D>([&](auto dc) {        // pairs behind this one still in flight: min(D - 1, NP - 1 - m)
                constexpr int d = decltype(dc)::value;
                if ((NP - 1 - m < D - 1 ? NP - 1 - m : D - 1) == d) x3_wait_vmcnt<8 * d>();
            });
            pair_fence();
            run_pair<AHEAD>(ring + (m % RING) * PAIRB, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                const int kb = 4 * pp + 2 * s + (i >> 2);
                mma3_w(acc1[i & 3][0], acc1[i & 3][1], wh, wl, ah[0][kb], al[0][kb], ah[1][kb], al[1][kb]);
            }, [&](int s, int q) { if (m + D < NP) kv_issue<E, RING>(sl, ring, wrsrc, wkv_off, wid, m + D, s, q); });
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
                if (plane_elems) {
                    // 24-bit rows (decoder_attn.h F24): the value rounded to 16 significant bits, bits 31..16 to the u16 plane at `dst`, bits
                    // 15..8 to the u8 plane `plane_elems` elements (2 bytes each) behind it — 16 + 8 bytes per lane instead of 32
                    const size_t at = (((size_t)image * heads + 2 * cc + pr) * 128 + token) * 32 + 8 * g;
                    unsigned w[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        w[r] = __float_as_uint(acc1[2 * pr][j][r] + bp0[32 * pr + r]) + 0x80u;
                        w[4 + r] = __float_as_uint(acc1[2 * pr + 1][j][r] + bp0[32 * pr + 4 + r]) + 0x80u;
                    }
                    u32x4 hi; uint2 lo;
                    hi[0] = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u); hi[1] = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
                    hi[2] = __builtin_amdgcn_perm(w[5], w[4], 0x07060302u); hi[3] = __builtin_amdgcn_perm(w[7], w[6], 0x07060302u);
                    lo.x = __builtin_amdgcn_perm(__builtin_amdgcn_perm(w[3], w[2], 0x05010501u), __builtin_amdgcn_perm(w[1], w[0], 0x05010501u), 0x05040100u);
                    lo.y = __builtin_amdgcn_perm(__builtin_amdgcn_perm(w[7], w[6], 0x05010501u), __builtin_amdgcn_perm(w[5], w[4], 0x05010501u), 0x05040100u);
                    unsigned char* hp = reinterpret_cast<unsigned char*>(dst);
                    *reinterpret_cast<u32x4*>(hp + at * 2) = hi;
                    *reinterpret_cast<uint2*>(hp + plane_elems * 2 + at) = lo;
                    continue;
                }
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
    size_t plane_elems = 0;      // != 0: K and V leave as 24-bit rows (u16 plane at kmem / vmem, u8 plane plane_elems elements behind it)
};

// wpack: the block-planar hi | lo copy of the f32 master (`wbytes` = 4 bytes per master element); pbase: the f32 master (vectors)
template <int E>
__global__ __launch_bounds__(256, 1)
void enc_blocks_x3_kernel(float* __restrict__ x, const unsigned char* __restrict__ wpack, unsigned wbytes, const float* __restrict__ pbase,
                          const EncBlockParams* __restrict__ blocks, int depth, float eps, int M, float* __restrict__ scratch, const EncTailX3 tail,
                          const EncHeadX3 head) {
    constexpr int F = 4 * E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    unsigned char* img = smem + IMG_OFF;
    float* sph = reinterpret_cast<float*>(smem + HEADS_PARAM_OFF);      // head loop (and proj's bias)
    float* sp = reinterpret_cast<float*>(smem + MLP_PARAM_OFF);        // MLP, tail

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const StreamLaneX sl(lane, wid, E);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wpack), 0, wbytes, 0x00020000);

    f32x4 acc[E / 16][2];
    bf16x8 ah[2][E / 32], al[2][E / 32];
    if (head.images) patch_head_x3<E>(head, ring, wrsrc, wid, lane, blockIdx.x, acc);
    else load_x_to_acc<E>(x, m0, M, wid, rr, g, acc);
    float* xbuf = scratch + (size_t)blockIdx.x * (2 * 48 * 1024);      // 48 pieces x 256 lanes x 4 floats: the parked residual stream
    float* obuf = xbuf + 48 * 1024;                                     // ... and the attention output fragments

    for (int l = 0; l < depth; ++l) {
        const EncBlockParams* bp = blocks + l;
        // ---- attention branch, head loop: parameters bqkv (3E) | bproj (E) | ln1 gamma (E) | ln1 beta (E)
        __syncthreads();
        heads_prefetch<E>(sl, ring, wrsrc, bp->wqkv, wid);
        params_to_lds(sph, pbase + bp->bqkv, 3 * E, tid);
        params_to_lds(sph + 3 * E, pbase + bp->bproj, E, tid);
        params_to_lds(sph + 4 * E, pbase + bp->ln1_w, E, tid);
        params_to_lds(sph + 5 * E, pbase + bp->ln1_b, E, tid);
        __syncthreads();
        ln_acc_to_frag<E>(acc, sph + 4 * E, sph + 5 * E, eps, g, ah, al);
        park_acc<E>(acc, xbuf, tid);
        heads_phase<E, X3_AHEAD>(ring, img, sph, wrsrc, bp->wqkv, 0.125f, sl, wid, tid, ah, al, obuf);
        // ---- attention branch, proj: x and the O fragments come back (each lane re-reads what it wrote)
        __syncthreads();                                                // every wave is done with the K / V^T images and the ring
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
        proj_phase<E, X3_MLP_RING, X3_AHEAD>(ring, wrsrc, bp->wproj, sl, wid, ah, al, acc);
        add_bias_to_acc<E>(sph + 3 * E, g, acc);
        // ---- MLP branch: parameters b1 (4E) | b2 (E) | ln2 gamma (E) | ln2 beta (E)
        __syncthreads();
        mlp3_prefetch<E>(sl, ring, wrsrc, bp->w1, bp->w2, wid);
        params_to_lds(sp, pbase + bp->b1, F, tid);
        params_to_lds(sp + F, pbase + bp->b2, E, tid);
        params_to_lds(sp + F + E, pbase + bp->ln2_w, E, tid);
        params_to_lds(sp + F + 2 * E, pbase + bp->ln2_b, E, tid);
        __syncthreads();
        ln_acc_to_frag<E>(acc, sp + F + E, sp + F + 2 * E, eps, g, ah, al);
        mlp_phase3<E, X3_AHEAD>(ring, sp, wrsrc, bp->w1, bp->w2, sl, wid, ah, al, acc);
        add_bias_to_acc<E>(sp + F, g, acc);
    }
    if (tail.kmem == nullptr) {
        store_acc_to_x<E>(x, m0, M, wid, rr, g, acc);
        return;
    }
    // ---- tail: parameters bkv (2E) | final norm gamma (E) | beta (E)
    __syncthreads();
    kv_prefetch<E, X3_MLP_RING>(sl, ring, wrsrc, tail.wkv, wid);
    params_to_lds(sp, pbase + tail.bkv, 2 * E, tid);
    params_to_lds(sp + 2 * E, pbase + tail.norm_w, E, tid);
    params_to_lds(sp + 3 * E, pbase + tail.norm_b, E, tid);
    __syncthreads();
    ln_acc_to_frag<E>(acc, sp + 2 * E, sp + 3 * E, eps, g, ah, al);
    kv_phase<E, X3_MLP_RING, X3_AHEAD>(ring, sp, wrsrc, tail.wkv, sl, wid, blockIdx.x, tail.heads, tail.kmem, tail.vmem, tail.plane_elems, ah, al);
}

template <int E>
hipError_t launch_enc_blocks_x3(hipStream_t s, float* x, const void* wpack, size_t wbytes, const float* pbase, const EncBlockParams* blocks,
                                       int depth, float eps, int M, float* scratch, const EncTailX3& tail = EncTailX3{0, 0, 0, 0, nullptr, nullptr, 0},
                                       const EncHeadX3& head = EncHeadX3{nullptr, 0, 0, nullptr}) {
    constexpr size_t lds = enc_blocks_x3_lds<E>();
    if (wbytes >= ((size_t)1 << 32) || M % 128 != 0) return hipErrorInvalidValue;
    auto kern = enc_blocks_x3_kernel<E>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(M / 128), dim3(256), lds, s, x, reinterpret_cast<const unsigned char*>(wpack), (unsigned)wbytes, pbase, blocks, depth, eps, M, scratch, tail, head);
    return hipGetLastError();
}

// Compiled in its own translation unit (kern_enc_blocks_x3.hip defines PQ_INSTANTIATE_ENC_BLOCKS_X3); every other unit only calls it.
#ifdef PQ_INSTANTIATE_ENC_BLOCKS_X3
template hipError_t launch_enc_blocks_x3<384>(hipStream_t, float*, const void*, size_t, const float*, const EncBlockParams*, int, float, int, float*, const EncTailX3&,
                                              const EncHeadX3&);
#else
extern template hipError_t launch_enc_blocks_x3<384>(hipStream_t, float*, const void*, size_t, const float*, const EncBlockParams*, int, float, int, float*, const EncTailX3&,
                                              const EncHeadX3&);
#endif

}  // namespace x3
}  // namespace pq
