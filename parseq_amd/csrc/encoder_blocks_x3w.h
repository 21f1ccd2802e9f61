// The one-launch bf16x3 encoder of encoder_blocks_x3.h re-cut for TWO waves per SIMD: one workgroup = one image = 128 rows as before,
// but EIGHT waves of 16 rows (one MFMA row tile each) at 256 registers instead of four waves of 32 rows at 512.
//
// Why.  With one 512-register wave per SIMD the kernel's time is the SUM of its matrix-pipe time (3.55 ms at batch 512) and of everything
// else the wave issues — LDS-DMA pieces, fragment reads, hi | lo splits, the exact-erf GELU, soft-max, LayerNorms, waits and barriers
// (≈ 3.9 ms; profiles/r04_x3_encoder_variants.md): an in-order wave cannot run its own VALU stretch under its own MFMAs.  With two waves on
// a SIMD the partner's MFMAs fill part of those stretches (tools/microbench/two_wave.hip: an MFMA wave and a VALU wave on one SIMD run side by
// side, the VALU wave at ~6.4 clk per independent instruction).  Measured on this kernel: 6.96 ms against 7.40 ms (median of 15 interleaved
// launches, batch 512, same box; profiles/r05_x3w_encoder.md) — the two waves run the SAME program between the SAME barriers, so a wave's GELU
// block still finds only its partner's share of one group of MFMAs to hide under.  What was tried on top and did not pay (asymmetric GELU
// placement with static priorities, a step-major GELU, a producer / consumer split of the MLP phase — half the waves fc1 + GELU, half fc2) is
// recorded in profiles/r05_x3w_encoder.md (the lab copy of this kernel that carried their switches was removed in round 6: `git show 186cd8e:tools/microbench/x3w_lab.h`).
// The price of 16 rows per wave is LDS traffic — a (hi, lo) weight-fragment pair feeds THREE MFMAs instead of six: 2 KiB per 48 clk per
// SIMD = 170 B/clk of the LDS's 256 B/clk for ds_read_b128 — and a register budget of 256 per lane: x (96) + the LayerNorm'd (hi, lo)
// operand (96) leave 64 for the accumulators of the running chunk, weight fragments and addresses.
//
// What is the same as encoder_blocks_x3.h — deliberately, so that the results are BIT-IDENTICAL to it (every accumulator receives the
// same products in the same order; tests/test_hip_ops.py compares the two kernels bit for bit):
//   * the arithmetic (bf16 pairs, three MFMAs per product, small terms first; fp32 LayerNorm / soft-max / erf GELU / residual);
//   * the block-planar hi | lo weight pack, the swizzled 16 KiB stage layout, the row orders, the K / V^T image planes, the LDS map;
//   * the phase structure: head loop (q | k | v pairs of stages through a five-slot ring, S / soft-max / P V from the image planes,
//     O to the workgroup's scratch), proj (pairs), MLP (triples, the two GELU blocks at group boundaries), tail (pairs);
//   * wave w of this kernel owns the rows that row tile (w & 1) of wave (w >> 1) owns there.
// What differs:
//   * a wave copies TWO 1-KiB pieces of every stage (rows 16 w + 8 q + (lane >> 3), q = 0, 1) instead of four;
//   * the unit is compiled in the MFMA VGPR form (below);
//   * the soft-max streams P: a 32-key k-block of P is exponentiated, split and consumed by its P V MFMAs before the next one.
#pragma once
#include "encoder_blocks_x3.h"

namespace pq {
namespace x3w {

using x3::STAGE; using x3::PAIRB; using x3::TRIPB; using x3::KIMG_B; using x3::VIMG_B; using x3::HEADS_SLOTS; using x3::IMG_OFF;
using x3::HEADS_PARAM_OFF; using x3::MLP_PARAM_OFF; using x3::split8; using x3::StreamLaneX; using x3::EncHeadX3; using x3::EncTailX3;

constexpr int AHEAD8 = 2;       // weight-fragment positions read ahead of the MFMAs (1: 7.22 ms, 2: 6.96 ms; profiles/r05_x3w_encoder.md)
constexpr int PARK8 = 4;        // accumulator tiles (of 24) that leave the register file for the head loop (4 .. 12 within 1 % of each other once proj reads the attention output just in time; fewest bytes)
constexpr int OSLOTS = 6;       // k-blocks of the attention output that proj keeps in registers at a time (of 12)
constexpr int NT = 512;        // threads of the workgroup: eight waves
// The kernel unit is compiled with -mllvm -amdgpu-mfma-vgpr-form (parseq_amd/build.py): accumulators live in VGPRs, one 256-register file per wave instead of a
// 128 | 128 VGPR / AGPR partition that neither the LayerNorm'd operand (96 + fragments + temporaries) nor the residual stream (96 + chunk accumulators) fits
// (in AGPR form the same source spills 183 registers, 35 - 48 in this form, and the kernel is 4 % slower).
#define PQ_X3W_ACC_PIN(x) asm volatile("" : "+v"(x))
#define PQ_X3W_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
// c += W A^T for ONE row tile, (hi, lo) fragments, small terms first — the order of x3::mma3_w per accumulator
__device__ __forceinline__ void mma3_w(f32x4& c, const bf16x8& wh, const bf16x8& wl, const bf16x8& ah, const bf16x8& al) {
    c = PQ_X3W_MFMA(wl, ah, c); c = PQ_X3W_MFMA(wh, al, c); c = PQ_X3W_MFMA(wh, ah, c);
}
// weights as the second operand (the v chunk: V^T) — x3::mma3_a
__device__ __forceinline__ void mma3_a(f32x4& c, const bf16x8& wh, const bf16x8& wl, const bf16x8& ah, const bf16x8& al) {
    c = PQ_X3W_MFMA(al, wh, c); c = PQ_X3W_MFMA(ah, wl, c); c = PQ_X3W_MFMA(ah, wh, c);
}

// Per-lane DMA source offsets of wave w (bytes of the block-planar pack): x3::StreamLaneX of wave w >> 1, moved four source rows on
// for the odd wave (its two pieces are pieces 2 and 3 of that wave's four: source rows + {4, 20}, LDS + 2048).
struct StreamLane8 {
    unsigned v64_, v128_, v128w_;
    __device__ __forceinline__ StreamLane8(int lane, int w8, int E) {
        const int wid = w8 >> 1, odd = w8 & 1;
        v64_ = StreamLaneX::calc<0>(lane, wid, E) + (unsigned)(odd * 4 * 4 * E);
        v128_ = StreamLaneX::calc<1>(lane, wid, E) + (unsigned)(odd * 4 * 4 * E);
        v128w_ = StreamLaneX::calc<2>(lane, wid, E) + (unsigned)(odd * 4 * 16 * E);
    }
    template <int KIND> __device__ __forceinline__ unsigned voff() const { return KIND == 0 ? v64_ : (KIND == 1 ? v128_ : v128w_); }
};
// the wave's two 1-KiB pieces of a stage: origin_b = byte offset of (row 0, k-block 0) of the stage in the pack, pitch_b = row pitch in bytes,
// dst = the stage's LDS slot + 2048 w.  (One M0 value: the second piece's immediate moves the LDS destination and is taken back out of the
// memory address through the scalar offset — StreamLane::issue_v.)
__device__ __forceinline__ void issue_stage(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned origin_b, unsigned pitch_b, unsigned char* dst) {
    auto* l = (__attribute__((address_space(3))) void*)dst;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin_b, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin_b + 16u * pitch_b - 1024u, 1024, 0);
}

__device__ __forceinline__ void group_fence() {         // the group about to run has landed (caller waited vmcnt); all waves are past the previous one
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_dma() { wait_vmcnt<N>(); }

// ---- NS stages under one barrier -------------------------------------------------------------------------------------------------
// stage_ptr(s): LDS address of stage s; mma(s, i, wh, wl): the three MFMAs that consume the (hi, lo) weight fragments of tile i (16 LDS rows)
// of stage s; issue(s): the wave's LDS-DMA pieces due at the start of stage s.  Fragment reads run AHEAD positions ahead of the MFMAs.
template <int NS, int AHEAD, class Ptr, class Mma, class Issue>
__device__ __forceinline__ void run_stages(Ptr&& stage_ptr, Mma&& mma, Issue&& issue) {
    const int ln = opaque_lane();
    const int fo0 = stage_frag_off(ln), fo1 = fo0 ^ 64;
    constexpr int NB = AHEAD + 1, NPOS = 8 * NS;
    bf16x8 wh[NB], wl[NB];
    static_for<0, AHEAD>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        const unsigned char* src = stage_ptr(n >> 3) + (n & 7) * 2048;
        wh[n] = *reinterpret_cast<const bf16x8*>(src + fo0); wl[n] = *reinterpret_cast<const bf16x8*>(src + fo1);
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NPOS>([&](auto nc) {
        constexpr int n = decltype(nc)::value, s = n >> 3, i = n & 7, nn = n + AHEAD;
        if constexpr (i == 0) { issue(s); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (nn < NPOS) {
            const unsigned char* src = stage_ptr(nn >> 3) + (nn & 7) * 2048;
            wh[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo0);
            wl[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo1);
        }
        mma(s, i, wh[n % NB], wl[n % NB]);
        if constexpr (nn < NPOS) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- x <-> accumulators (one row tile: encoder_blocks.h load_x_to_acc / store_acc_to_x / add_bias_to_acc for rows 16 w + r16) ----------
template <int E>
__device__ __forceinline__ void load_x_to_acc(const float* __restrict__ x, int m0, int M, int w8, int rr, int g, f32x4 (&acc)[E / 16]) {
    constexpr int KSTEPS = E / 32;
    const bool lo_half = rr < 8;
    const int rbase = m0 + w8 * 16 + (rr & 7);
    const float* xlo = x + (size_t)min(rbase, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
    const float* xhi = x + (size_t)min(rbase + 8, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const u32x4 p0 = *reinterpret_cast<const u32x4*>(xlo + ks * 32);        // a piece of row (r16 & 7)
        const u32x4 p1 = *reinterpret_cast<const u32x4*>(xhi + ks * 32);        // a piece of row (r16 & 7) + 8
        const u32x4 got = swap_half_rows(lo_half ? p1 : p0);
        const u32x4 ev = lo_half ? p0 : got, od = lo_half ? got : p1;
        acc[(ks >> 2) * 8 + 2 * (ks & 3)] = f32x4{__uint_as_float(ev[0]), __uint_as_float(ev[1]), __uint_as_float(ev[2]), __uint_as_float(ev[3])};
        acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1] = f32x4{__uint_as_float(od[0]), __uint_as_float(od[1]), __uint_as_float(od[2]), __uint_as_float(od[3])};
    }
}
template <int E>
__device__ __forceinline__ void store_acc_to_x(float* __restrict__ x, int m0, int M, int w8, int rr, int g, const f32x4 (&acc)[E / 16]) {
    const bool lo_half = rr < 8;
    const int r_first = m0 + w8 * 16 + (rr & 7), r_second = r_first + 8;
    const int cbase = 8 * g + (lo_half ? 0 : 4);
#pragma unroll
    for (int q32 = 0; q32 < E / 32; ++q32) {
        const f32x4 ta = acc[(q32 >> 2) * 8 + 2 * (q32 & 3)], tb = acc[(q32 >> 2) * 8 + 2 * (q32 & 3) + 1];
        const u32x4 pa = {__float_as_uint(ta[0]), __float_as_uint(ta[1]), __float_as_uint(ta[2]), __float_as_uint(ta[3])};
        const u32x4 pb = {__float_as_uint(tb[0]), __float_as_uint(tb[1]), __float_as_uint(tb[2]), __float_as_uint(tb[3])};
        const u32x4 got = swap_half_rows(lo_half ? pb : pa);
        const u32x4 first = lo_half ? pa : got, second = lo_half ? got : pb;
        const int col = 32 * q32 + cbase;
        if (r_first < M) *reinterpret_cast<u32x4*>(x + (size_t)r_first * E + col) = first;
        if (r_second < M) *reinterpret_cast<u32x4*>(x + (size_t)r_second * E + col) = second;
    }
}
template <int E>
__device__ __forceinline__ void add_bias_to_acc(const float* sb, int g, f32x4 (&acc)[E / 16]) {
#pragma unroll
    for (int q32 = 0; q32 < E / 32; ++q32) {
        const float4 b0 = *reinterpret_cast<const float4*>(sb + 32 * q32 + 8 * g), b1 = *reinterpret_cast<const float4*>(sb + 32 * q32 + 8 * g + 4);
        f32x4& ta = acc[(q32 >> 2) * 8 + 2 * (q32 & 3)];
        f32x4& tb = acc[(q32 >> 2) * 8 + 2 * (q32 & 3) + 1];
        ta[0] += b0.x; ta[1] += b0.y; ta[2] += b0.z; ta[3] += b0.w;
        tb[0] += b1.x; tb[1] += b1.y; tb[2] += b1.z; tb[3] += b1.w;
    }
}
__device__ __forceinline__ void params_to_lds(float* dst, const float* __restrict__ src, int n, int tid) {
    for (int i = tid; i < n; i += NT) dst[i] = src[i];
}

// LayerNorm of the rows held in the accumulators -> (hi, lo) operand fragments (x3::ln_acc_to_frag for one row tile: same passes, same order)
template <int E>
__device__ __forceinline__ void ln_acc_to_frag(const f32x4 (&acc)[E / 16], const float* sgam, const float* sbet, float eps, int g,
                                               bf16x8 (&ah)[E / 32], bf16x8 (&al)[E / 32]) {
    constexpr int KSTEPS = E / 32;
    auto read8 = [&](int ks, float (&x)[8]) {
        const f32x4& a = acc[(ks >> 2) * 8 + 2 * (ks & 3)]; const f32x4& b = acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1];
        // (every value through an empty asm, once per pass: without it the three passes share ONE copy of the 96 accumulator values, the allocator answers with twice the spills,
        // and the kernel is 1.8 % slower than with the 288 extra moves — profiles/r05_x3w_encoder.md)
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[r] = a[r]; x[4 + r] = b[r]; asm volatile("" : "+v"(x[r]), "+v"(x[4 + r])); }
    };
    float s1 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        float x[8];
        read8(ks, x);
        s1 += x3::ln_sum8(x);
    }
    s1 = rows4_sum(s1);
    const float mean = s1 * (1.0f / E);
    float s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        float x[8];
        read8(ks, x);
        s2 += x3::ln_sq8(x, mean);
    }
    s2 = rows4_sum(s2);
    const float rstd = x3::ln_rstd(s2, E, eps);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        float x[8];
        read8(ks, x);
        const float4 ga = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g + 4);
        const float4 ba = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g + 4);
        float v[8];
        x3::ln_norm8(x, mean, rstd, ga, gb, ba, bb, v);
        split8(v, ah[ks], al[ks]);
    }
}

// A lane's element of a piece-major scratch buffer: wave-uniform base (SGPR pair) + the lane's 32-bit byte offset (ONE register, 16 tid) — the global_load / _store saddr +
// voffset form.  As 64-bit per-lane pointers these addresses are register pairs the allocator spills and reloads inside the MFMA streams (each reload with an s_waitcnt vmcnt(0)).
template <class T> __device__ __forceinline__ T* lane_at(T* uniform_base, unsigned lane_bytes) {
    return reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(const_cast<std::remove_const_t<T>*>(uniform_base)) + lane_bytes);
}
// ---- the residual stream's first PARK8 tiles leave the register file for the head loop (piece-major across the 512 lanes) ----
template <int E>
__device__ __forceinline__ void park_acc(const f32x4 (&acc)[E / 16], float* __restrict__ dst, unsigned lb) {
#pragma unroll
    for (int i = 0; i < PARK8; ++i) *reinterpret_cast<f32x4*>(lane_at(dst + (size_t)i * NT * 4, lb)) = acc[i];
}
template <int E>
__device__ __forceinline__ void unpark_acc(f32x4 (&acc)[E / 16], const float* __restrict__ src, unsigned lb) {
#pragma unroll
    for (int i = 0; i < PARK8; ++i) acc[i] = *reinterpret_cast<const f32x4*>(lane_at(src + (size_t)i * NT * 4, lb));
}

// ---- head loop (x3::heads_phase for one row tile per wave) -----------------------------------------------------------------------------
// O of head h, k-block pr: pieces (2 h + pr) * 2 + {0 hi, 1 lo} of `obuf` (piece-major across the 512 lanes): the k-block 2 h + pr operand of proj.
template <int E>
__device__ __forceinline__ void heads_issue(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int w8, int sigma) {
    const int m = sigma >> 1, s = sigma & 1, h = m / 9, n = m - 9 * h;
    unsigned char* dst = ring + (sigma % HEADS_SLOTS) * STAGE + w8 * 2048;
    const int u = n / 3, pp = n - 3 * u, t = 2 * pp + s;
    issue_stage(wrsrc, sl.template voff<0>(), (wqkv_off + (unsigned)((u * E + h * 64) * E + t * 64)) * 4u, 4u * E, dst);
}
template <int E>
__device__ __forceinline__ void heads_prefetch(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int w8) {
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, w8, 0);
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, w8, 1);
    heads_issue<E>(sl, ring, wrsrc, wqkv_off, w8, 2);
}

template <int E, int AHEAD>
__device__ __forceinline__ void heads_phase(unsigned char* ring, unsigned char* img, const float* sbq, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off,
                                            float scale, const StreamLane8& sl, int w8, int tid, const bf16x8 (&ah)[E / 32],
                                            const bf16x8 (&al)[E / 32], float* __restrict__ obuf) {
    constexpr int H = E / 64;
    static_assert(E == 384, "written for E = 384");
    unsigned char* kimg_h = img; unsigned char* kimg_l = img + KIMG_B;
    unsigned char* vimg_h = img + 2 * KIMG_B; unsigned char* vimg_l = vimg_h + VIMG_B;
    const float sc2 = scale * 1.44269504088896340736f;

    for (int h = 0; h < H; ++h) {
        f32x4 acc1[4];
        bf16x8 qh[2], ql[2];
        static_for<0, 9>([&](auto nc) {
            constexpr int n = decltype(nc)::value, u = n / 3, pp = n % 3;
            const int m = 9 * h + n;                             // pair index of the phase
            // this pair has landed; the stage after it (two pieces per wave) may be in flight
            if (m + 1 < 9 * H) wait_dma<2>(); else wait_dma<0>();
            group_fence();
            if constexpr (pp == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            auto issue = [&](int s) { if (2 * m + 3 + s < 18 * H) heads_issue<E>(sl, ring, wrsrc, wqkv_off, w8, 2 * m + 3 + s); };
            const unsigned char* st0 = ring + ((2 * m) % HEADS_SLOTS) * STAGE;
            const unsigned char* st1 = ring + ((2 * m + 1) % HEADS_SLOTS) * STAGE;
            auto sptr = [&](int s) { return s ? st1 : st0; };
            if constexpr (u < 2) {
                run_stages<2, AHEAD>(sptr, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 4 * pp + 2 * s + (i >> 2);
                    mma3_w(acc1[i & 3], wh, wl, ah[kb], al[kb]);
                }, issue);
            } else {
                run_stages<2, AHEAD>(sptr, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 4 * pp + 2 * s + (i >> 2);
                    mma3_a(acc1[i & 3], wh, wl, ah[kb], al[kb]);
                }, issue);
            }
            if constexpr (u < 2 && pp == 2) {
                // q -> fragments, k -> the K image planes (rows in the order the P fragments need: encoder_attn_fused.h)
                const int ln = opaque_lane();
                const int rr = ln & 15, g = ln >> 4;
                const int krow = 32 * (w8 >> 1) + 16 * ((rr >> 2) & 1) + 4 * (rr >> 3) + (rr & 3) + 8 * (w8 & 1);
                const float* bp0 = sbq + u * E + h * 64 + 8 * g;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc1[2 * pr][r] + bp0[32 * pr + r];
                        v[4 + r] = acc1[2 * pr + 1][r] + bp0[32 * pr + 4 + r];
                    }
                    bf16x8 fh, fl;
                    split8(v, fh, fl);
                    if constexpr (u == 0) { qh[pr] = fh; ql[pr] = fl; }
                    else {
                        const int off = krow * AF_KROWB + 64 * pr + 16 * g;
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
                    bf16x4 fh, fl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc1[i][r] + bv;
                        fh[r] = static_cast<bf16_t>(v);
                        fl[r] = static_cast<bf16_t>(v - static_cast<float>(fh[r]));
                    }
                    const int off = (16 * i + rr) * AF_VROWB + 2 * (16 * w8 + 4 * g);
                    *reinterpret_cast<bf16x4*>(vimg_h + off) = fh;
                    *reinterpret_cast<bf16x4*>(vimg_l + off) = fl;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // S^T = K Q^T for the wave's 16 queries: 16 positions (k-step ks = n / 8, key tile kt = n % 8), image fragments AHEAD positions ahead
                f32x4 sc[8];
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                {
                    constexpr int NB = AHEAD + 1;
                    bf16x8 fh[NB], fl[NB];
                    const int base = rr * AF_KROWB + 16 * g;
                    static_for<0, AHEAD>([&](auto pc) {
                        constexpr int p = decltype(pc)::value, off = 16 * (p & 7) * AF_KROWB + 64 * (p >> 3);
                        fh[p] = *reinterpret_cast<const bf16x8*>(kimg_h + base + off); fl[p] = *reinterpret_cast<const bf16x8*>(kimg_l + base + off);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, 16>([&](auto pc) {
                        constexpr int p = decltype(pc)::value, ks = p >> 3, kt = p & 7, pn = p + AHEAD;
                        if constexpr (pn < 16) {
                            constexpr int off = 16 * (pn & 7) * AF_KROWB + 64 * (pn >> 3);
                            fh[pn % NB] = *reinterpret_cast<const bf16x8*>(kimg_h + base + off); fl[pn % NB] = *reinterpret_cast<const bf16x8*>(kimg_l + base + off);
                        }
                        mma3_w(sc[kt], fh[p % NB], fl[p % NB], qh[ks], ql[ks]);
                        if constexpr (pn < 16) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        } else __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
                mx = rows4_max(mx);
                const float mc = mx * sc2;
                float sum = 0.f;
                f32x4 ov[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) ov[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int vbase = rr * AF_VROWB + 16 * g;
                // P streamed: k-block ks of P (32 keys) is exponentiated and split, then consumed by its four P V tiles (the accumulation order per O tile — ks ascending — is x3's)
                static_for<0, 4>([&](auto kc) {
                    constexpr int ks = decltype(kc)::value;
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = __builtin_amdgcn_exp2f(sc[2 * ks][r] * sc2 - mc);
                        v[4 + r] = __builtin_amdgcn_exp2f(sc[2 * ks + 1][r] * sc2 - mc);
                        sum += v[r] + v[4 + r];
                    }
                    bf16x8 ph, pl;
                    split8(v, ph, pl);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const int off = vbase + 16 * dt * AF_VROWB + 64 * ks;
                        const bf16x8 fh = *reinterpret_cast<const bf16x8*>(vimg_h + off), fl = *reinterpret_cast<const bf16x8*>(vimg_l + off);
                        mma3_w(ov[dt], fh, fl, ph, pl);
                    }
                });
                sum = rows4_sum(sum);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = ov[2 * pr][r] * inv; v[4 + r] = ov[2 * pr + 1][r] * inv; }
                    bf16x8 fh, fl;
                    split8(v, fh, fl);
                    float* o = obuf + (size_t)(((2 * h + pr) * 2) * NT) * 4;
                    *reinterpret_cast<bf16x8*>(lane_at(o, (unsigned)tid * 16u)) = fh;
                    *reinterpret_cast<bf16x8*>(lane_at(o + NT * 4, (unsigned)tid * 16u)) = fl;
                }
            }
        });
    }
}

// ---- proj: acc2 += Wproj O  (bias NOT added): 36 stages of 128 rows x one k-block, stage t = (k-block t / 3, row group t % 3), 18 pairs ----
template <int E, int RING>
__device__ __forceinline__ void proj_issue(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, int w8, int n, int s) {
    const int t = 2 * n + s, kb = t / 3, ng = t - 3 * kb;
    unsigned char* dst = ring + (n % RING) * PAIRB + s * STAGE + w8 * 2048;
    issue_stage(wrsrc, sl.template voff<1>(), (wproj_off + (unsigned)(ng * 128 * E + kb * 32)) * 4u, 4u * E, dst);
}
template <int E, int RING>
__device__ __forceinline__ void proj_prefetch(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, int w8) {
    static_for<0, RING - 1>([&](auto nc) {
        proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, w8, decltype(nc)::value, 0);
        proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, w8, decltype(nc)::value, 1);
    });
}
// The attention output is read back from the workgroup's scratch JUST IN TIME: only OSLOTS = 6 of its 12 k-blocks are in registers at a time — k-block kb + 6's (hi, lo)
// fragments are loaded into k-block kb's registers at the first pair boundary behind kb's last stage (3 kb + 2; first use seven pairs on).  The loads are older than every
// LDS-DMA piece issued from there on, so the pair boundaries' counted vmcnt waits cover them.  48 registers fewer in the phase that holds all of x again: no scratch reload
// (with its s_waitcnt vmcnt(0), which drains the weight stream) is left inside the stream (-3 %, profiles/r05_x3w_encoder.md).
template <int E, int RING, int AHEAD>
__device__ __forceinline__ void proj_phase(unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, const StreamLane8& sl, int w8,
                                           bf16x8 (&oh)[E / 32], bf16x8 (&ol)[E / 32], f32x4 (&acc2)[E / 16], const float* oback, unsigned lb) {
    constexpr int NP = 3 * (E / 32) / 2, D = RING - 1;
    static_assert(E == 384 && OSLOTS == 6, "written for E = 384");
    static_for<0, NP>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int behind = (NP - 1 - n) < (D - 1) ? (NP - 1 - n) : (D - 1);       // pairs issued after this one and still in flight
        wait_dma<4 * behind>();
        group_fence();
        static_for<0, OSLOTS>([&](auto kc) {
            constexpr int kb = decltype(kc)::value;
            if constexpr ((3 * kb + 2) / 2 + 1 == n) {
                const float* o = oback + (size_t)(((kb + OSLOTS) * 2) * NT) * 4;
                oh[kb] = *reinterpret_cast<const bf16x8*>(lane_at(o, lb));
                ol[kb] = *reinterpret_cast<const bf16x8*>(lane_at(o + NT * 4, lb));
            }
        });
        const unsigned char* grp = ring + (n % RING) * PAIRB;
        run_stages<2, AHEAD>([&](int s) { return grp + s * STAGE; }, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
            const int t = 2 * n + s, kb = t / 3, ng = t % 3;
            mma3_w(acc2[ng * 8 + i], wh, wl, oh[kb % OSLOTS], ol[kb % OSLOTS]);
        }, [&](int s) { if constexpr (n + D < NP) proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, w8, n + D, s); });
    });
}

// ---- MLP phase: acc2 += fc2(gelu(fc1(a) + b1))  (bias of fc2 NOT added), triples of stages (x3::mlp_phase3) ---------------------------
// GELU of hidden units 32 pr + [0, 32) of the chunk -> (hi, lo) fragments
__device__ __forceinline__ void gelu_frag(const f32x4 (&acc1)[4], const float* bp, int pr, bf16x8& hh, bf16x8& hl) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = gelu_erf(acc1[2 * pr][q] + bp[32 * pr + q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[4 + q] = gelu_erf(acc1[2 * pr + 1][q] + bp[32 * pr + 4 + q]);
    split8(v, hh, hl);
}
// triple k of chunk c (k = 0, 1: fc1 stages 3 k .. 3 k + 2; k = 2, 3: fc2 k-block k - 2, row groups 0 .. 2), stage s, into ring group (4 c + k) % 3
template <int E>
__device__ __forceinline__ void mlp_issue(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          int w8, int c, int k, int s) {
    constexpr int F = 4 * E;
    unsigned char* dst = ring + ((4 * c + k) % 3) * TRIPB + s * STAGE + w8 * 2048;
    if (k < 2) issue_stage(wrsrc, sl.template voff<0>(), (w1_off + (unsigned)(c * 64 * E + (3 * k + s) * 64)) * 4u, 4u * E, dst);
    else issue_stage(wrsrc, sl.template voff<2>(), (w2_off + (unsigned)(s * 128 * F + c * 64 + (k - 2) * 32)) * 4u, 4u * F, dst);
}
template <int E>
__device__ __forceinline__ void mlp_prefetch(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off, int w8) {
    static_for<0, 2>([&](auto kc) {
        static_for<0, 3>([&](auto sc) { mlp_issue<E>(sl, ring, wrsrc, w1_off, w2_off, w8, 0, decltype(kc)::value, decltype(sc)::value); });
    });
}
template <int E, int AHEAD>
__device__ __forceinline__ void mlp_phase(unsigned char* ring, const float* sb1, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          const StreamLane8& sl, int w8, const bf16x8 (&ah)[E / 32], const bf16x8 (&al)[E / 32], f32x4 (&acc2)[E / 16]) {
    constexpr int F = 4 * E, NCH = F / 64;
    static_assert(E == 384, "written for E = 384");
    for (int c = 0; c < NCH; ++c) {
        const bool last = c + 1 == NCH;
        f32x4 acc1[4];
        bf16x8 hh, hl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            PQ_X3W_ACC_PIN(acc1[i]);
        }
        const int g = opaque_lane() >> 4;
        const float* bp = sb1 + c * 64 + 8 * g;
        static_for<0, 4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            // in flight behind this triple: the next one (six pieces per wave) — none behind the phase's last
            if (!last || k < 3) wait_dma<6>(); else wait_dma<0>();
            group_fence();
            if constexpr (k >= 2) gelu_frag(acc1, bp, k - 2, hh, hl);      // hidden units 32 (k - 2) .. + 32 of the chunk: this triple's k-block
            auto issue = [&](int s) {      // the triple two ahead
                if constexpr (k < 2) mlp_issue<E>(sl, ring, wrsrc, w1_off, w2_off, w8, c, k + 2, s);
                else if (!last) mlp_issue<E>(sl, ring, wrsrc, w1_off, w2_off, w8, c + 1, k - 2, s);
            };
            const unsigned char* grp = ring + ((4 * c + k) % 3) * TRIPB;
            auto sptr = [&](int s) { return grp + s * STAGE; };
            if constexpr (k < 2) {
                run_stages<3, AHEAD>(sptr, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    const int kb = 2 * (3 * k + s) + (i >> 2);
                    mma3_w(acc1[i & 3], wh, wl, ah[kb], al[kb]);
                }, issue);
            } else {
                run_stages<3, AHEAD>(sptr, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    mma3_w(acc2[s * 8 + i], wh, wl, hh, hl);
                }, issue);
            }
        });
    }
}

// ---- head: x = patches W_pe^T + (pos_embed + bias)  (x3::patch_head_x3 for one row tile per wave) ---------------------------------------
template <int E>
__device__ __forceinline__ void patch_head(const EncHeadX3& hp, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, int w8, int lane, int image, f32x4 (&acc)[E / 16]) {
    static_assert(E == 384 && 9 * STAGE <= (int)x3::enc_blocks_x3_lds<384>(), "three 128-row groups; nine stages fit the launch's LDS");
    constexpr int PK = 96, IH = 32, IW = 128;
    const int rr = lane & 15, g = lane >> 4;
    const unsigned vpe = StreamLaneX::calc<1>(lane, w8 >> 1, PK) + (unsigned)((w8 & 1) * 4 * 4 * PK);
    static_for<0, 9>([&](auto sc) {
        constexpr int st = decltype(sc)::value, ng = st / 3, kb = st % 3;
        issue_stage(wrsrc, vpe, (hp.wpe + (unsigned)(ng * 128 * PK + kb * 32)) * 4u, 4u * PK, ring + st * STAGE + w8 * 2048);
    });
    load_x_to_acc<E>(hp.posb, 0, 128, w8, rr, g, acc);
    bf16x8 ph[3], pl[3];
    const int token = 16 * w8 + rr, gy = token >> 4, gx = token & 15;
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
        split8(v, ph[c], pl[c]);
    }
    wait_vmcnt<0>();
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
            mma3_w(acc[ng * 8 + i], wh, wl, ph[kb], pl[kb]);
        }
    });
}

// ---- tail: K | V = LayerNorm_final(x) Wkv^T + bkv  (x3::kv_phase for one row tile per wave) ----------------------------------------------
template <int E, int RING>
__device__ __forceinline__ void kv_issue(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, int w8, int m, int s) {
    const int c = m / 3, pp = m - 3 * c;      // pair m = 3 c + pp of the tail (group m % RING, issued RING - 1 pairs ahead), stage s
    unsigned char* dst = ring + (m % RING) * PAIRB + s * STAGE + w8 * 2048;
    issue_stage(wrsrc, sl.template voff<0>(), (wkv_off + (unsigned)(c * 64 * E + (2 * pp + s) * 64)) * 4u, 4u * E, dst);
}
template <int E, int RING>
__device__ __forceinline__ void kv_prefetch(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, int w8) {
    static_for<0, RING - 1>([&](auto mc) {
        kv_issue<E, RING>(sl, ring, wrsrc, wkv_off, w8, decltype(mc)::value, 0);
        kv_issue<E, RING>(sl, ring, wrsrc, wkv_off, w8, decltype(mc)::value, 1);
    });
}
template <int E, int RING, int AHEAD>
__device__ __forceinline__ void kv_phase(unsigned char* ring, const float* sbkv, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, const StreamLane8& sl,
                                         int w8, int image, int heads, float* __restrict__ kmem, float* __restrict__ vmem, size_t plane_elems,
                                         const bf16x8 (&ah)[E / 32], const bf16x8 (&al)[E / 32]) {
    constexpr int NC = 2 * E / 64, NP = 3 * NC, D = RING - 1;
    for (int c = 0; c < NC; ++c) {
        f32x4 acc1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<0, 3>([&](auto pc) {
            constexpr int pp = decltype(pc)::value;
            const int m = 3 * c + pp;
            static_for<0, D>([&](auto dc) {        // pairs behind this one still in flight: min(D - 1, NP - 1 - m)
                constexpr int d = decltype(dc)::value;
                if ((NP - 1 - m < D - 1 ? NP - 1 - m : D - 1) == d) wait_dma<4 * d>();
            });
            group_fence();
            const unsigned char* grp = ring + (m % RING) * PAIRB;
            run_stages<2, AHEAD>([&](int s) { return grp + s * STAGE; }, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                const int kb = 4 * pp + 2 * s + (i >> 2);
                mma3_w(acc1[i & 3], wh, wl, ah[kb], al[kb]);
            }, [&](int s) { if (m + D < NP) kv_issue<E, RING>(sl, ring, wrsrc, wkv_off, w8, m + D, s); });
        });
        const int ln = opaque_lane();
        const int rr = ln & 15, g = ln >> 4;
        float* dst = c < NC / 2 ? kmem : vmem;
        const int cc = c < NC / 2 ? c : c - NC / 2;
        const float* bp0 = sbkv + c * 64 + 8 * g;
        const int token = 16 * w8 + rr;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            if (plane_elems) {
                // 24-bit rows (decoder_attn.h F24): bits 31..16 of the value rounded to 16 significant bits to the u16 plane at `dst`, bits 15..8 to the u8 plane behind it
                const size_t at = (((size_t)image * heads + 2 * cc + pr) * 128 + token) * 32 + 8 * g;
                unsigned w[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    w[r] = __float_as_uint(acc1[2 * pr][r] + bp0[32 * pr + r]) + 0x80u;
                    w[4 + r] = __float_as_uint(acc1[2 * pr + 1][r] + bp0[32 * pr + 4 + r]) + 0x80u;
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
            *reinterpret_cast<float4*>(o) = make_float4(acc1[2 * pr][0] + bp0[32 * pr], acc1[2 * pr][1] + bp0[32 * pr + 1],
                                                        acc1[2 * pr][2] + bp0[32 * pr + 2], acc1[2 * pr][3] + bp0[32 * pr + 3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(acc1[2 * pr + 1][0] + bp0[32 * pr + 4], acc1[2 * pr + 1][1] + bp0[32 * pr + 5],
                                                            acc1[2 * pr + 1][2] + bp0[32 * pr + 6], acc1[2 * pr + 1][3] + bp0[32 * pr + 7]);
        }
    }
}

template <int E>
__global__ __launch_bounds__(NT, 1)
void enc_blocks_x3w_kernel(float* __restrict__ x, const unsigned char* __restrict__ wpack, unsigned wbytes, const float* __restrict__ pbase,
                           const EncBlockParams* __restrict__ blocks, int depth, float eps, int M, float* __restrict__ scratch, const EncTailX3 tail,
                           const EncHeadX3 head) {
    constexpr int F = 4 * E, RING = X3_MLP_RING;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    unsigned char* img = smem + IMG_OFF;
    float* sph = reinterpret_cast<float*>(smem + HEADS_PARAM_OFF);      // head loop (and proj's bias)
    float* sp = reinterpret_cast<float*>(smem + MLP_PARAM_OFF);        // MLP, tail

    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const StreamLane8 sl(lane, w8, E);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wpack), 0, wbytes, 0x00020000);

    f32x4 acc[E / 16];
    bf16x8 ah[E / 32], al[E / 32];
    if (head.images) patch_head<E>(head, ring, wrsrc, w8, lane, blockIdx.x, acc);
    else load_x_to_acc<E>(x, m0, M, w8, rr, g, acc);
    float* xbuf = scratch + (size_t)blockIdx.x * (2 * 48 * 1024);      // the parked tiles of the residual stream (PARK8 x 512 lanes x 4 floats)
    float* obuf = xbuf + 48 * 1024;                                     // ... and the attention output fragments (24 pieces x 512 lanes x 4 floats)

    for (int l = 0; l < depth; ++l) {
        const EncBlockParams* bp = blocks + l;
        {
            // ---- attention branch, head loop: parameters bqkv (3E) | bproj (E) | ln1 gamma (E) | ln1 beta (E)
            __syncthreads();
            heads_prefetch<E>(sl, ring, wrsrc, bp->wqkv, w8);
            params_to_lds(sph, pbase + bp->bqkv, 3 * E, tid);
            params_to_lds(sph + 3 * E, pbase + bp->bproj, E, tid);
            params_to_lds(sph + 4 * E, pbase + bp->ln1_w, E, tid);
            params_to_lds(sph + 5 * E, pbase + bp->ln1_b, E, tid);
            __syncthreads();
            ln_acc_to_frag<E>(acc, sph + 4 * E, sph + 5 * E, eps, g, ah, al);
            park_acc<E>(acc, xbuf, (unsigned)tid * 16u);
            heads_phase<E, AHEAD8>(ring, img, sph, wrsrc, bp->wqkv, 0.125f, sl, w8, tid, ah, al, obuf);
            // ---- attention branch, proj: x and the O fragments come back (each lane re-reads what it wrote)
            __syncthreads();                                                // every wave is done with the K / V^T images and the ring
            // whatever the compiler itself moved out of the register file for the head loop comes back HERE, while no LDS-DMA is in flight
#pragma unroll
            for (int i = PARK8; i < E / 16; ++i) PQ_X3W_ACC_PIN(acc[i]);
            // (the addresses go through an empty asm: the optimiser must not forward the stored values to these loads; the lane's byte offset too, so that it is a live
            // register — not a scratch reload — behind the prefetch)
            const float* xback = xbuf; const float* oback = obuf;
            unsigned lb = (unsigned)tid * 16u;
            asm volatile("" : "+s"(xback), "+s"(oback), "+v"(lb) :: "memory");
            proj_prefetch<E, RING>(sl, ring, wrsrc, bp->wproj, w8);
            unpark_acc<E>(acc, xback, lb);
#pragma unroll
            for (int kb = 0; kb < OSLOTS; ++kb) {
                const float* o = oback + (size_t)((kb * 2) * NT) * 4;
                ah[kb] = *reinterpret_cast<const bf16x8*>(lane_at(o, lb));
                al[kb] = *reinterpret_cast<const bf16x8*>(lane_at(o + NT * 4, lb));
            }
            proj_phase<E, RING, AHEAD8>(ring, wrsrc, bp->wproj, sl, w8, ah, al, acc, oback, lb);
            add_bias_to_acc<E>(sph + 3 * E, g, acc);
        }
        {
            // ---- MLP branch: parameters b1 (4E) | b2 (E) | ln2 gamma (E) | ln2 beta (E)
            __syncthreads();
            mlp_prefetch<E>(sl, ring, wrsrc, bp->w1, bp->w2, w8);
            params_to_lds(sp, pbase + bp->b1, F, tid);
            params_to_lds(sp + F, pbase + bp->b2, E, tid);
            params_to_lds(sp + F + E, pbase + bp->ln2_w, E, tid);
            params_to_lds(sp + F + 2 * E, pbase + bp->ln2_b, E, tid);
            __syncthreads();
            ln_acc_to_frag<E>(acc, sp + F + E, sp + F + 2 * E, eps, g, ah, al);
            mlp_phase<E, AHEAD8>(ring, sp, wrsrc, bp->w1, bp->w2, sl, w8, ah, al, acc);
            add_bias_to_acc<E>(sp + F, g, acc);
        }
    }
    if (tail.kmem == nullptr) {
        store_acc_to_x<E>(x, m0, M, w8, rr, g, acc);
        return;
    }
    // ---- tail: parameters bkv (2E) | final norm gamma (E) | beta (E)
    __syncthreads();
    kv_prefetch<E, RING>(sl, ring, wrsrc, tail.wkv, w8);
    params_to_lds(sp, pbase + tail.bkv, 2 * E, tid);
    params_to_lds(sp + 2 * E, pbase + tail.norm_w, E, tid);
    params_to_lds(sp + 3 * E, pbase + tail.norm_b, E, tid);
    __syncthreads();
    ln_acc_to_frag<E>(acc, sp + 2 * E, sp + 3 * E, eps, g, ah, al);
    kv_phase<E, RING, AHEAD8>(ring, sp, wrsrc, tail.wkv, sl, w8, blockIdx.x, tail.heads, tail.kmem, tail.vmem, tail.plane_elems, ah, al);
}

template <int E>
hipError_t launch_enc_blocks_x3w(hipStream_t s, float* x, const void* wpack, size_t wbytes, const float* pbase, const EncBlockParams* blocks,
                                 int depth, float eps, int M, float* scratch, const EncTailX3& tail = EncTailX3{0, 0, 0, 0, nullptr, nullptr, 0},
                                 const EncHeadX3& head = EncHeadX3{nullptr, 0, 0, nullptr}) {
    constexpr size_t lds = x3::enc_blocks_x3_lds<E>();
    if (wbytes >= ((size_t)1 << 32) || M % 128 != 0) return hipErrorInvalidValue;
    auto kern = enc_blocks_x3w_kernel<E>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(M / 128), dim3(NT), lds, s, x, reinterpret_cast<const unsigned char*>(wpack), (unsigned)wbytes, pbase, blocks, depth, eps, M, scratch, tail, head);
    return hipGetLastError();
}

// Compiled in its own translation unit (kern_enc_blocks_x3w.hip defines PQ_INSTANTIATE_ENC_BLOCKS_X3W, with the VGPR-form flag); every other unit only calls it.
#ifdef PQ_INSTANTIATE_ENC_BLOCKS_X3W
template hipError_t launch_enc_blocks_x3w<384>(hipStream_t, float*, const void*, size_t, const float*, const EncBlockParams*, int, float, int, float*, const EncTailX3&,
                                               const EncHeadX3&);
#else
extern template hipError_t launch_enc_blocks_x3w<384>(hipStream_t, float*, const void*, size_t, const float*, const EncBlockParams*, int, float, int, float*, const EncTailX3&,
                                                      const EncHeadX3&);
#endif

}  // namespace x3w
}  // namespace pq
