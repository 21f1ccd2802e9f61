// LAB COPY (tools/microbench): parseq_amd/csrc/encoder_blocks_x3w.h with every experiment of round 5 still in it — role-form MLP phase, step-major GELU, ping-pong GELU placement,
// static priority, timing ablations (X3W_ABLATE), in-kernel phase timers (X3W_TIMERS).  tools/x3_variants.sh builds it (names starting with w8), tools/x3_variant_bench.py times the builds.
// The one-launch bf16x3 encoder of encoder_blocks_x3.h re-cut for TWO waves per SIMD: one workgroup = one image = 128 rows as before,
// but EIGHT waves of 16 rows (one MFMA row tile each) at 256 registers instead of four waves of 32 rows at 512.
//
// Why.  With one 512-register wave per SIMD the kernel's time is the SUM of its matrix-pipe time (3.55 ms at batch 512) and of everything
// else the wave issues — LDS-DMA pieces, fragment reads, hi | lo splits, the exact-erf GELU, soft-max, LayerNorms, waits and barriers
// (≈ 3.9 ms; profiles/r04_x3_encoder_variants.md): an in-order wave cannot run its own VALU stretch under its own MFMAs.  Two waves per
// SIMD can: while one wave is in a GELU / soft-max / LayerNorm / split stretch, its partner's MFMAs keep the SIMD's matrix pipe busy.
// The price is LDS traffic — a (hi, lo) weight-fragment pair now feeds THREE MFMAs (one row tile) instead of six — which at
// 2 KiB per 48 clk per SIMD is 170 B/clk of the LDS's 256 B/clk (ds_read_b128, MI355X_MICROARCH.md) — and a register budget of 256 per
// lane: x (96) + the LayerNorm'd (hi, lo) operand (96) leave 64 for accumulators of the running chunk, weight fragments and addresses.
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
//   * fragment reads run ONE position ahead through two (hi, lo) buffers (the partner wave hides the rest of the LDS latency);
//   * the MLP phase places its GELU blocks asymmetrically: waves 0-3 ("half A", static priority 1) evaluate a GELU block right AFTER
//     the MFMAs of the group in front of it, waves 4-7 ("half B", the SIMD partners) right BEFORE the MFMAs of the group behind it, so
//     that on every SIMD one wave's GELU runs under the other wave's MFMAs instead of both evaluating it at the same time;
//   * the soft-max streams P: a 32-key k-block of P is exponentiated, split and consumed by its P V MFMAs before the next one.
#pragma once
#include "x3_lab.h"

namespace pq {
namespace x3w {

using x3::STAGE; using x3::PAIRB; using x3::TRIPB; using x3::KIMG_B; using x3::VIMG_B; using x3::HEADS_SLOTS; using x3::IMG_OFF;
using x3::HEADS_PARAM_OFF; using x3::MLP_PARAM_OFF; using x3::split8; using x3::StreamLaneX; using x3::EncHeadX3; using x3::EncTailX3;

#ifndef X3W_AHEAD
#define X3W_AHEAD 1            // weight-fragment positions read ahead of the MFMAs
#endif
#ifndef X3W_PARK_TILES
#define X3W_PARK_TILES 8       // accumulator tiles (of 24) that leave the register file for the head loop
#endif
#ifndef X3W_PINGPONG
#define X3W_PINGPONG 1         // 1: asymmetric GELU placement in the MLP phase (half A after its group, half B before the next); 0: both halves before the next group
#endif
#ifndef X3W_GELU_STEPMAJOR
#define X3W_GELU_STEPMAJOR 1   // 1: the GELU block evaluated step-major over its four element pairs (gelu8_split), 0: the scalar form as the compiler schedules it (pair by pair)
#endif
#ifndef X3W_VGPR_FORM
#define X3W_VGPR_FORM 0        // 1: the unit is compiled with -mllvm -amdgpu-mfma-vgpr-form (accumulators in VGPRs: one 256-register file, no VGPR / AGPR partition)
#endif
#ifndef X3W_MLP_ROLES
#define X3W_MLP_ROLES 1        // 1: the MLP phase in role form (mlp_roles: half A fc1 + GELU, half B fc2), 0: in lock-step form (mlp_phase)
#endif
#ifndef X3W_RL_AHEAD
#define X3W_RL_AHEAD 2         // weight-fragment positions read ahead of the MFMAs in the role-form MLP phase
#endif
#ifndef X3W_PRIO
#define X3W_PRIO 1             // 1: half A runs at s_setprio 1 for the whole kernel
#endif
// Timing ablations (tools/x3_variants.sh builds only; results are WRONG with any bit set): 1 GELU -> identity, 4 no LDS-DMA issue, 16 no group barriers,
// 32 no waits for the LDS-DMA, 64 one MFMA per product, 128 no weight-fragment reads (the registers keep what they held), 256 no exp in the soft-max,
// 512 role form: half A skips its stages, 1024 role form: half B skips its stages
#ifndef X3W_ABLATE
#define X3W_ABLATE 0
#endif
constexpr int NT = 512;        // threads of the workgroup: eight waves
// X3W_TIMERS=1 (variant builds only): every wave accumulates s_memtime ticks per phase and stores them over its rows of x (tools/x3_variant_bench.py --timers8).
// Slots: 0 parameters + LayerNorm, 1 park / unpark / O reload, 2 head loop groups (wait + barrier), 3 head loop MFMA pairs, 4 head loop epilogues + soft-max section,
// 5 proj wait + barrier, 6 proj MFMAs, 7 MLP wait + barrier, 8 MLP GELU in front of a group (half B), 9 MLP fc1 MFMAs, 10 MLP fc2 MFMAs, 11 MLP GELU behind a group (half A), 12 tail + rest
#ifndef X3W_TIMERS
#define X3W_TIMERS 0
#endif
#if X3W_TIMERS
struct Timers { long long acc[13]; long long last; };
#define X3W_TICK(slot) do { const long long now_ = clock64(); x3t->acc[slot] += now_ - x3t->last; x3t->last = now_; } while (0)
#define X3W_TARG , Timers* x3t
#define X3W_TPASS , x3t
#else
#define X3W_TICK(slot)
#define X3W_TARG
#define X3W_TPASS
#endif

__device__ __forceinline__ float x3w_gelu(float x) { if constexpr ((X3W_ABLATE & 1) != 0) return x; else return gelu_erf(x); }
__device__ __forceinline__ float x3w_exp2(float x) { if constexpr ((X3W_ABLATE & 256) != 0) return x; else return __builtin_amdgcn_exp2f(x); }
#if X3W_VGPR_FORM
__device__ __forceinline__ float acc_read(const float& a) { float v = a; asm volatile("" : "+v"(v)); return v; }
#define PQ_X3W_ACC_PIN(x) asm volatile("" : "+v"(x))
#else
using x3::acc_read;
#define PQ_X3W_ACC_PIN(x) asm volatile("" : "+a"(x))
#endif
#define PQ_X3W_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
// c += W A^T for ONE row tile, (hi, lo) fragments, small terms first — the order of x3::mma3_w per accumulator
__device__ __forceinline__ void mma3_w(f32x4& c, const bf16x8& wh, const bf16x8& wl, const bf16x8& ah, const bf16x8& al) {
    if constexpr ((X3W_ABLATE & 64) == 0) { c = PQ_X3W_MFMA(wl, ah, c); c = PQ_X3W_MFMA(wh, al, c); }
    c = PQ_X3W_MFMA(wh, ah, c);
}
// weights as the second operand (the v chunk: V^T) — x3::mma3_a
__device__ __forceinline__ void mma3_a(f32x4& c, const bf16x8& wh, const bf16x8& wl, const bf16x8& ah, const bf16x8& al) {
    if constexpr ((X3W_ABLATE & 64) == 0) { c = PQ_X3W_MFMA(al, wh, c); c = PQ_X3W_MFMA(ah, wl, c); }
    c = PQ_X3W_MFMA(ah, wh, c);
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
    if constexpr ((X3W_ABLATE & 4) != 0) return;
    auto* l = (__attribute__((address_space(3))) void*)dst;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin_b, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin_b + 16u * pitch_b - 1024u, 1024, 0);
}

#ifndef X3W_DMA_HALF
#define X3W_DMA_HALF 0         // 1: only waves 0-3 issue LDS-DMA pieces — their own two of a stage and their partner's (wave + 4) two; 2: only waves 4-7
#endif
// KIND: the stage's row order / pitch combination (StreamLaneX::calc) — it fixes where wave w + 4's pieces lie relative to wave w's: 64 LDS rows on,
// i.e. the second k-block of the same source rows (KIND 0: + 128 bytes) or 64 source rows on (KIND 1, 2: + 64 row pitches)
template <int KIND>
__device__ __forceinline__ void issue_stage_k(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned origin_b, unsigned pitch_b, unsigned char* dst, int w8) {
    if constexpr (X3W_DMA_HALF == 0) issue_stage(rsrc, voff, origin_b, pitch_b, dst);
    else {
        // the issuing half's lanes hold the voff of THEIR wave; the other half's pieces differ by a wave-uniform delta
        const unsigned delta = KIND == 0 ? 128u : 64u * pitch_b;
        if (X3W_DMA_HALF == 1 ? w8 < 4 : w8 >= 4) {
            issue_stage(rsrc, voff, origin_b, pitch_b, dst);
            if (X3W_DMA_HALF == 1) issue_stage(rsrc, voff, origin_b + delta, pitch_b, dst + 8192);
            else issue_stage(rsrc, voff, origin_b - delta, pitch_b, dst - 8192);
        }
    }
}
__device__ __forceinline__ void group_fence() {         // the group about to run has landed (caller waited vmcnt); all waves are past the previous one
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr ((X3W_ABLATE & 16) == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_dma() { if constexpr ((X3W_ABLATE & 32) == 0) wait_vmcnt<(X3W_DMA_HALF ? 2 * N : N)>(); }

// X3W_PRIO 4 / 5: waves 0-3 take priority 1 in even (half) stages, waves 4-7 in odd ones (the wave's half is read from the hardware wave id register through
// s_getreg: no live register needed)
__device__ __forceinline__ void g_prio_toggle(int s) {
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    if (((s ^ w) & 1) == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
}
// ---- NS stages under one barrier -------------------------------------------------------------------------------------------------
// stage_ptr(s): LDS address of stage s; mma(s, i, wh, wl): the three MFMAs that consume the (hi, lo) weight fragments of tile i (16 LDS rows)
// of stage s; issue(s): the wave's LDS-DMA pieces due at the start of stage s.  Fragment reads run AHEAD positions ahead of the MFMAs.
#ifndef X3W_FO_LIVE
#define X3W_FO_LIVE 0          // 1: the per-lane LDS fragment offsets of run_stages come from threadIdx (loop-invariant to the compiler: two live registers) instead of being
                               // recomputed at every call through opaque_lane() (~10 VALU instructions x ~170 calls per block)
#endif
struct NoFill { template <int N> __device__ __forceinline__ void at() const {} static constexpr int per_pos = 0; };
template <int NS, int AHEAD, class Ptr, class Mma, class Issue, class Fill = NoFill>
__device__ __forceinline__ void run_stages(Ptr&& stage_ptr, Mma&& mma, Issue&& issue, Fill&& fill = Fill{}) {
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
        if constexpr (i == 0) {
            if constexpr (X3W_PRIO == 4 || X3W_PRIO == 5) {          // alternate the two halves' priority stage by stage (4) / every half stage (5): approximates fair sharing of the SIMD
                g_prio_toggle(s);
            }
            issue(s); __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i == 4 && X3W_PRIO == 5) g_prio_toggle(s + 1);
        if constexpr (nn < NPOS && (X3W_ABLATE & 128) == 0) {
            const unsigned char* src = stage_ptr(nn >> 3) + (nn & 7) * 2048;
            wh[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo0);
            wl[nn % NB] = *reinterpret_cast<const bf16x8*>(src + fo1);
        }
        if constexpr ((X3W_ABLATE & 128) != 0) asm volatile("" : "+v"(wh[n % NB]), "+v"(wl[n % NB]));
        fill.template at<n>();          // a slice of VALU work that rides between this position's MFMAs (NoFill: nothing)
        mma(s, i, wh[n % NB], wl[n % NB]);
        constexpr int FP = std::remove_reference_t<Fill>::per_pos;      // VALU instructions of the slice, spread behind the three MFMAs
        if constexpr (nn < NPOS) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (FP > 0) __builtin_amdgcn_sched_group_barrier(0x002, (FP + 2) / 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (FP > 0) __builtin_amdgcn_sched_group_barrier(0x002, (FP + 1) / 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (FP > 0) __builtin_amdgcn_sched_group_barrier(0x002, FP / 3, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (FP > 0) __builtin_amdgcn_sched_group_barrier(0x002, (FP + 2) / 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (FP > 0) __builtin_amdgcn_sched_group_barrier(0x002, (FP + 1) / 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (FP > 0) __builtin_amdgcn_sched_group_barrier(0x002, FP / 3, 0);
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
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[r] = acc_read(a[r]); x[4 + r] = acc_read(b[r]); }
    };
    float s1 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        float x[8];
        read8(ks, x);
        s1 += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    }
    s1 = rows4_sum(s1);
    const float mean = s1 * (1.0f / E);
    float s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        float x[8];
        read8(ks, x);
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
        read8(ks, x);
        const float4 ga = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g + 4);
        const float4 ba = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g + 4);
        float v[8];
        v[0] = (x[0] - mean) * rstd * ga.x + ba.x; v[1] = (x[1] - mean) * rstd * ga.y + ba.y;
        v[2] = (x[2] - mean) * rstd * ga.z + ba.z; v[3] = (x[3] - mean) * rstd * ga.w + ba.w;
        v[4] = (x[4] - mean) * rstd * gb.x + bb.x; v[5] = (x[5] - mean) * rstd * gb.y + bb.y;
        v[6] = (x[6] - mean) * rstd * gb.z + bb.z; v[7] = (x[7] - mean) * rstd * gb.w + bb.w;
        split8(v, ah[ks], al[ks]);
    }
}

// ---- the residual stream's first X3W_PARK_TILES tiles leave the register file for the head loop (piece-major across the 512 lanes) ----
template <int E>
__device__ __forceinline__ void park_acc(const f32x4 (&acc)[E / 16], float* __restrict__ dst, int tid) {
#pragma unroll
    for (int i = 0; i < X3W_PARK_TILES; ++i) *reinterpret_cast<f32x4*>(dst + ((size_t)i * NT + tid) * 4) = acc[i];
}
template <int E>
__device__ __forceinline__ void unpark_acc(f32x4 (&acc)[E / 16], const float* __restrict__ src, int tid) {
#pragma unroll
    for (int i = 0; i < X3W_PARK_TILES; ++i) acc[i] = *reinterpret_cast<const f32x4*>(src + ((size_t)i * NT + tid) * 4);
}

// ---- head loop (x3::heads_phase for one row tile per wave) -----------------------------------------------------------------------------
// O of head h, k-block pr: pieces (2 h + pr) * 2 + {0 hi, 1 lo} of `obuf` (piece-major across the 512 lanes): the k-block 2 h + pr operand of proj.
template <int E>
__device__ __forceinline__ void heads_issue(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int w8, int sigma) {
    const int m = sigma >> 1, s = sigma & 1, h = m / 9, n = m - 9 * h;
    unsigned char* dst = ring + (sigma % HEADS_SLOTS) * STAGE + w8 * 2048;
    const int u = n / 3, pp = n - 3 * u, t = 2 * pp + s;
    issue_stage_k<0>(wrsrc, sl.template voff<0>(), (wqkv_off + (unsigned)((u * E + h * 64) * E + t * 64)) * 4u, 4u * E, dst, w8);
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
                                            const bf16x8 (&al)[E / 32], float* __restrict__ obuf X3W_TARG) {
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
            X3W_TICK(2);
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
            X3W_TICK(3);
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
                        v[r] = x3w_exp2(sc[2 * ks][r] * sc2 - mc);
                        v[4 + r] = x3w_exp2(sc[2 * ks + 1][r] * sc2 - mc);
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
                    float* o = obuf + ((size_t)(((2 * h + pr) * 2) * NT) + tid) * 4;
                    *reinterpret_cast<bf16x8*>(o) = fh;
                    *reinterpret_cast<bf16x8*>(o + NT * 4) = fl;
                }
            }
            X3W_TICK(4);
        });
    }
}

// ---- proj: acc2 += Wproj O  (bias NOT added): 36 stages of 128 rows x one k-block, stage t = (k-block t / 3, row group t % 3), 18 pairs ----
template <int E, int RING>
__device__ __forceinline__ void proj_issue(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, int w8, int n, int s) {
    const int t = 2 * n + s, kb = t / 3, ng = t - 3 * kb;
    unsigned char* dst = ring + (n % RING) * PAIRB + s * STAGE + w8 * 2048;
    issue_stage_k<1>(wrsrc, sl.template voff<1>(), (wproj_off + (unsigned)(ng * 128 * E + kb * 32)) * 4u, 4u * E, dst, w8);
}
template <int E, int RING>
__device__ __forceinline__ void proj_prefetch(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, int w8) {
    static_for<0, RING - 1>([&](auto nc) {
        proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, w8, decltype(nc)::value, 0);
        proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, w8, decltype(nc)::value, 1);
    });
}
#ifndef X3W_O_JIT
#define X3W_O_JIT 0            // 1: proj keeps only SIX k-blocks of the attention output in registers: k-block kb + 6's (hi, lo) fragments are loaded from the workgroup's scratch into
                               // k-block kb's registers at the first pair boundary behind kb's last stage (48 registers fewer in the phase that holds all of x again)
#endif
template <int E, int RING, int AHEAD>
__device__ __forceinline__ void proj_phase(unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wproj_off, const StreamLane8& sl, int w8,
                                           bf16x8 (&oh)[E / 32], bf16x8 (&ol)[E / 32], f32x4 (&acc2)[E / 16], const float* oback, int tid X3W_TARG) {
    constexpr int NP = 3 * (E / 32) / 2, D = RING - 1, OS = X3W_O_JIT ? 6 : E / 32;
    static_assert(E == 384, "written for E = 384");
    static_for<0, NP>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int behind = (NP - 1 - n) < (D - 1) ? (NP - 1 - n) : (D - 1);       // pairs issued after this one and still in flight
        wait_dma<4 * behind>();
        group_fence();
        X3W_TICK(5);
        if constexpr (X3W_O_JIT) {
            // k-block kb's last stage is 3 kb + 2, in pair (3 kb + 2) / 2: behind it its registers take k-block kb + 6 (first used in stage 3 kb + 18, seven pairs on; the loads are older
            // than every LDS-DMA piece issued from here on, so the pair boundaries' counted vmcnt waits cover them)
            static_for<0, 6>([&](auto kc) {
                constexpr int kb = decltype(kc)::value;
                if constexpr ((3 * kb + 2) / 2 + 1 == n) {
                    const float* o = oback + ((size_t)(((kb + 6) * 2) * NT) + tid) * 4;
                    oh[kb] = *reinterpret_cast<const bf16x8*>(o);
                    ol[kb] = *reinterpret_cast<const bf16x8*>(o + NT * 4);
                }
            });
        }
        const unsigned char* grp = ring + (n % RING) * PAIRB;
        run_stages<2, AHEAD>([&](int s) { return grp + s * STAGE; }, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
            const int t = 2 * n + s, kb = t / 3, ng = t % 3;
            mma3_w(acc2[ng * 8 + i], wh, wl, oh[kb % OS], ol[kb % OS]);
        }, [&](int s) { if constexpr (n + D < NP) proj_issue<E, RING>(sl, ring, wrsrc, wproj_off, w8, n + D, s); });
        X3W_TICK(6);
    });
}

// ---- MLP phase: acc2 += fc2(gelu(fc1(a) + b1))  (bias of fc2 NOT added), triples of stages (x3::mlp_phase3) ---------------------------
// GELU of hidden units 32 pr + [0, 32) of the chunk -> (hi, lo) fragments
// The arithmetic is common.h gelu_erf followed by x3::split8, operation for operation (including the two contractions the compiler makes there:
// r = fma(-e, p t, 1) and lo = bf16(fma(hx, 1 + erf, -hi))), so the results are bit-identical to x3::gelu_frag — but evaluated STEP-MAJOR over the four
// element pairs: every step is four independent instructions (the empty asm between steps pins the order).  Pair by pair, as the compiler schedules the
// scalar form, every instruction depends on the one before it; next to a partner wave that issues MFMAs a dependent VALU instruction costs ~10 clk against ~6.4
// for an independent one (tools/microbench/two_wave.hip), and the GELU block (~130 instructions) outlasts the partner's MFMA group it is supposed to hide under.
#define PQ_X3W_PIN4(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
__device__ __forceinline__ void gelu8_split(f32x2 (&v)[4], bf16x8& hi, bf16x8& lo) {
#pragma clang fp contract(off)
    if constexpr ((X3W_ABLATE & 1) != 0) {
        float y[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { y[2 * i] = v[i][0]; y[2 * i + 1] = v[i][1]; }
        split8(y, hi, lo);
        return;
    }
    f32x2 z[4], hx[4], t[4], e[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = v[i] * 0.70710678118654752440f;
#pragma unroll
    for (int i = 0; i < 4; ++i) hx[i] = v[i] * 0.5f;
    PQ_X3W_PIN4(z); PQ_X3W_PIN4(hx);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 az = __builtin_elementwise_abs(z[i]);
        t[i] = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, az, f32x2{1.0f, 1.0f});
    }
    PQ_X3W_PIN4(t);
#pragma unroll
    for (int i = 0; i < 4; ++i) { t[i][0] = __builtin_amdgcn_rcpf(t[i][0]); t[i][1] = __builtin_amdgcn_rcpf(t[i][1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 az = __builtin_elementwise_abs(z[i]);
        e[i] = (az * -az) * 1.44269502162933349609f;          // __expf(-ax * ax): the argument of v_exp_f32
    }
    PQ_X3W_PIN4(e); PQ_X3W_PIN4(t);
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[i][0] = __builtin_amdgcn_exp2f(e[i][0]); e[i][1] = __builtin_amdgcn_exp2f(e[i][1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t[i], f32x2{-1.453152027f, -1.453152027f});
    PQ_X3W_PIN4(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2{1.421413741f, 1.421413741f});
    PQ_X3W_PIN4(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2{-0.284496736f, -0.284496736f});
    PQ_X3W_PIN4(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2{0.254829592f, 0.254829592f});
    PQ_X3W_PIN4(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = t[i] * p[i];
    PQ_X3W_PIN4(p); PQ_X3W_PIN4(e);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(-e[i], p[i], f32x2{1.0f, 1.0f});          // r = 1 - p t e
    PQ_X3W_PIN4(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i][0] = __builtin_copysignf(p[i][0], z[i][0]); p[i][1] = __builtin_copysignf(p[i][1], z[i][1]); }
    PQ_X3W_PIN4(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = p[i] + 1.0f;                                                        // w = 1 + erf
    PQ_X3W_PIN4(p);
    f32x2 y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = hx[i] * p[i];
    PQ_X3W_PIN4(y);
#pragma unroll
    for (int i = 0; i < 4; ++i) { hi[2 * i] = static_cast<bf16_t>(y[i][0]); hi[2 * i + 1] = static_cast<bf16_t>(y[i][1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 hf = {static_cast<float>(hi[2 * i]), static_cast<float>(hi[2 * i + 1])};
        y[i] = __builtin_elementwise_fma(hx[i], p[i], -hf);
    }
    PQ_X3W_PIN4(y);
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[2 * i] = static_cast<bf16_t>(y[i][0]); lo[2 * i + 1] = static_cast<bf16_t>(y[i][1]); }
}
__device__ __forceinline__ void gelu_frag(const f32x4 (&acc1)[4], const float* bp, int pr, bf16x8& hh, bf16x8& hl) {
#if X3W_GELU_STEPMAJOR
    f32x2 v[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        v[q] = f32x2{acc1[2 * pr][2 * q] + bp[32 * pr + 2 * q], acc1[2 * pr][2 * q + 1] + bp[32 * pr + 2 * q + 1]};
        v[2 + q] = f32x2{acc1[2 * pr + 1][2 * q] + bp[32 * pr + 4 + 2 * q], acc1[2 * pr + 1][2 * q + 1] + bp[32 * pr + 4 + 2 * q + 1]};
    }
    gelu8_split(v, hh, hl);
#else
    float v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = x3w_gelu(acc1[2 * pr][q] + bp[32 * pr + q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[4 + q] = x3w_gelu(acc1[2 * pr + 1][q] + bp[32 * pr + 4 + q]);
    split8(v, hh, hl);
#endif
}
// X3W_GELU_FILL: the chunk's SECOND GELU block (hidden units 32 .. 63: complete when the second fc1 triple is) evaluated in 24 slices that ride between the MFMAs of the
// first fc2 triple (which consumes the first block's fragments), one element pair per six positions: the arithmetic of gelu8_split (bit-identical to gelu_erf + split8).
// With two waves on a SIMD a wave's own MFMAs are ~32 clk apart: up to two VALU instructions behind each cost nothing (tools/microbench/two_wave.hip).
#ifndef X3W_GELU_FILL
#define X3W_GELU_FILL 0
#endif
struct GeluFill {
    static constexpr int per_pos = 5;
    const f32x4 (&acc1)[4];
    const float* bp;
    bf16x8& hh; bf16x8& hl;
    f32x2 v, z, hx, t, e, p;
    __device__ __forceinline__ GeluFill(const f32x4 (&a)[4], const float* b, bf16x8& h, bf16x8& l) : acc1(a), bp(b), hh(h), hl(l) {}
    template <int N> __device__ __forceinline__ void at() {
#pragma clang fp contract(off)
        constexpr int pr = N / 6, st = N % 6;          // element pair 0..3 of the block: values (2 pr, 2 pr + 1) of the lane's eight
        if constexpr (N < 24) {
            // (every slice starts from values pinned by an empty volatile asm: without it instruction selection gathers the whole chain at its first use — sched_barrier only
            // binds the later machine scheduler)
            if constexpr (st == 0) {
                constexpr int tile = 2 + (pr >> 1), q = 2 * (pr & 1);
                float a0 = acc1[tile][q], a1 = acc1[tile][q + 1];
                asm volatile("" : "+v"(a0), "+v"(a1));
                v = f32x2{a0 + bp[32 + 4 * (pr >> 1) + q], a1 + bp[32 + 4 * (pr >> 1) + q + 1]};
                z = v * 0.70710678118654752440f;
                hx = v * 0.5f;
            } else if constexpr (st == 1) {
                asm volatile("" : "+v"(z));
                const f32x2 az = __builtin_elementwise_abs(z);
                t = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, az, f32x2{1.0f, 1.0f});
                t[0] = __builtin_amdgcn_rcpf(t[0]); t[1] = __builtin_amdgcn_rcpf(t[1]);
                e = (az * -az) * 1.44269502162933349609f;
            } else if constexpr (st == 2) {
                asm volatile("" : "+v"(e), "+v"(t));
                e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
                p = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
                p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
            } else if constexpr (st == 3) {
                asm volatile("" : "+v"(p), "+v"(t));
                p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
                p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
                p = t * p;
            } else if constexpr (st == 4) {
                asm volatile("" : "+v"(p), "+v"(e));
                p = __builtin_elementwise_fma(-e, p, f32x2{1.0f, 1.0f});
                p[0] = __builtin_copysignf(p[0], z[0]); p[1] = __builtin_copysignf(p[1], z[1]);
                p = p + 1.0f;
            } else {
                asm volatile("" : "+v"(p), "+v"(hx));
                const f32x2 y = hx * p;
                hh[2 * pr] = static_cast<bf16_t>(y[0]); hh[2 * pr + 1] = static_cast<bf16_t>(y[1]);
                const f32x2 hf = {static_cast<float>(hh[2 * pr]), static_cast<float>(hh[2 * pr + 1])};
                const f32x2 l = __builtin_elementwise_fma(hx, p, -hf);
                hl[2 * pr] = static_cast<bf16_t>(l[0]); hl[2 * pr + 1] = static_cast<bf16_t>(l[1]);
            }
        }
    }
};
// triple k of chunk c (k = 0, 1: fc1 stages 3 k .. 3 k + 2; k = 2, 3: fc2 k-block k - 2, row groups 0 .. 2), stage s, into ring group (4 c + k) % 3
template <int E>
__device__ __forceinline__ void mlp_issue(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          int w8, int c, int k, int s) {
    constexpr int F = 4 * E;
    unsigned char* dst = ring + ((4 * c + k) % 3) * TRIPB + s * STAGE + w8 * 2048;
    if (k < 2) issue_stage_k<0>(wrsrc, sl.template voff<0>(), (w1_off + (unsigned)(c * 64 * E + (3 * k + s) * 64)) * 4u, 4u * E, dst, w8);
    else issue_stage_k<2>(wrsrc, sl.template voff<2>(), (w2_off + (unsigned)(s * 128 * F + c * 64 + (k - 2) * 32)) * 4u, 4u * F, dst, w8);
}
template <int E>
__device__ __forceinline__ void mlp_prefetch(const StreamLane8& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off, int w8) {
    static_for<0, 2>([&](auto kc) {
        static_for<0, 3>([&](auto sc) { mlp_issue<E>(sl, ring, wrsrc, w1_off, w2_off, w8, 0, decltype(kc)::value, decltype(sc)::value); });
    });
}
template <int E, int AHEAD>
__device__ __forceinline__ void mlp_phase(unsigned char* ring, const float* sb1, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          const StreamLane8& sl, int w8, const bf16x8 (&ah)[E / 32], const bf16x8 (&al)[E / 32], f32x4 (&acc2)[E / 16] X3W_TARG) {
    constexpr int F = 4 * E, NCH = F / 64;
    static_assert(E == 384, "written for E = 384");
    const bool half_b = X3W_PINGPONG ? w8 >= 4 : true;      // half B (and everybody without the ping-pong): GELU in front of the group that consumes it
    for (int c = 0; c < NCH; ++c) {
        const bool last = c + 1 == NCH;
        f32x4 acc1[4];
        bf16x8 hh, hl;
#if X3W_GELU_FILL
        bf16x8 hh1, hl1;          // the second block's fragments, produced while the first block's are being consumed
#endif
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
            X3W_TICK(7);
#if X3W_GELU_FILL
            if constexpr (k == 2) gelu_frag(acc1, bp, 0, hh, hl);
#else
            if constexpr (k >= 2) { if (half_b) gelu_frag(acc1, bp, k - 2, hh, hl); }
#endif
            X3W_TICK(8);
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
                X3W_TICK(9);
            } else {
#if X3W_GELU_FILL
                if constexpr (k == 2) {
                    GeluFill gf(acc1, bp, hh1, hl1);
                    run_stages<3, AHEAD>(sptr, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                        mma3_w(acc2[s * 8 + i], wh, wl, hh, hl);
                    }, issue, gf);
                } else {
                    run_stages<3, AHEAD>(sptr, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                        mma3_w(acc2[s * 8 + i], wh, wl, hh1, hl1);
                    }, issue);
                }
#else
                run_stages<3, AHEAD>(sptr, [&](int s, int i, const bf16x8& wh, const bf16x8& wl) {
                    mma3_w(acc2[s * 8 + i], wh, wl, hh, hl);
                }, issue);
#endif
                X3W_TICK(10);
            }
#if !X3W_GELU_FILL
            if constexpr (k == 1 || k == 2) { if (!half_b) gelu_frag(acc1, bp, k - 1, hh, hl); }      // half A: behind its own MFMAs, under half B's
#endif
            X3W_TICK(11);
        });
    }
}

// ---- MLP phase in ROLE form --------------------------------------------------------------------------------------------------------------
// The lock-step form above cannot hide its GELU: both waves of a SIMD consume the same stages between the same barriers, so a wave's GELU block has only the
// partner's share of ONE group of MFMAs to hide under, and next to a partner that issues MFMAs a VALU instruction costs 6 - 15 clk (tools/microbench/two_wave.hip;
// in-kernel timers: the block of ~130 instructions takes 1 500 - 2 000 clk against 1 152 clk of MFMAs per group and wave).  And every weight fragment pair read from
// LDS feeds three MFMAs instead of six.  Role form: the two waves of a SIMD stop doing the same thing.
//   half A (waves 0-3)  owns the LayerNorm'd operand `a` of 32 rows (its own 16 and its SIMD partner's 16: 192 registers) and computes fc1 + GELU;
//   half B (waves 4-7)  owns the residual stream x of the same 32 rows (192 registers) and computes fc2 into it.
// Every weight fragment pair feeds six MFMAs again (two row tiles), A's GELU runs under B's MFMAs of a DIFFERENT part of the stream, and the hidden activations
// cross from A to B through a 16 KiB LDS buffer as ready-made (hi, lo) operand fragments.  The unit of work is 32 hidden units u (48 per block):
//   A   three stages of 32 W1 rows x four k-blocks (stage kind K32) accumulate acc1 (16 registers), then GELU + split of the unit's 16 values per lane -> h[u] in LDS;
//   B   reads h[u] (16 registers) and runs the three row-group stages of k-block u of W2 into x.
// One stage each per barrier interval t: A consumes its stage t, B its stage t - 4 (one unit + one interval behind: h[u] is written at the start of interval 3 u + 3
// and read at the start of interval 3 u + 4), through ONE ring of eight stages: interval t's pair of stages lives in slots 2 (t % 4) + {0 A, 1 B} and is copied
// three intervals ahead by all eight waves (two 1-KiB pieces per wave and stage; at the ends of the stream the missing stage of a pair is a clamped duplicate so that
// every interval carries four pieces per wave and the vmcnt immediates stay constant).  Around the pipeline the register files are exchanged through the ring's LDS:
// x of half A's rows goes to B, the LayerNorm'd fragments of half B's rows go to A, and x comes back at the end.  Accumulation order per accumulator is x3's.
constexpr int RL_P = 3, RL_SLOTS = 2 * (RL_P + 1);                 // intervals ahead, ring slots
constexpr int RL_HBUF_OFF = RL_SLOTS * STAGE;                        // 131072: h[u] fragments, 16 KiB
static_assert(RL_HBUF_OFF + 16384 <= MLP_PARAM_OFF, "ring | h | parameters");
#ifndef X3W_RL_PRIO
#define X3W_RL_PRIO 0          // 1: half A (the GELU half) at s_setprio 1 during the role-form MLP phase, 2: half B
#endif
// per-lane DMA offset of stage kind K32 (32 W1 rows x four k-blocks at pitch 4E bytes): wave w copies tile w of the stage — rows of pair member w & 1, k-block w >> 1
__device__ __forceinline__ unsigned voff_k32(int lane, int w8, int E) {
    const int l3 = lane >> 3;
    return (unsigned)(((l3 >> 2) * 8 + (l3 & 3) + 4 * (w8 & 1)) * 4 * E + (w8 >> 1) * 128 + ((lane & 7) ^ l3) * 16);
}
template <int E>
__device__ __forceinline__ void rl_issue(unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off, unsigned vk32, unsigned v128w, int w8, int t2) {
    constexpr int F = 4 * E, NU = F / 32;
    int ka = t2, kb = t2 - 4;
    ka = ka > 3 * NU - 1 ? 3 * NU - 1 : ka;
    kb = kb < 0 ? 0 : (kb > 3 * NU - 1 ? 3 * NU - 1 : kb);
    unsigned char* dst = ring + (t2 & 3) * (2 * STAGE) + w8 * 2048;
    const int ua = ka / 3, sa = ka - 3 * ua, ub = kb / 3, sb = kb - 3 * ub;
    issue_stage(wrsrc, vk32, (w1_off + (unsigned)(ua * 32 * E + sa * 128)) * 4u, 4u * E, dst);
    issue_stage(wrsrc, v128w, (w2_off + (unsigned)(sb * 128 * F + ub * 32)) * 4u, 4u * F, dst + STAGE);
}
// one stage, two row tiles per wave: mma(i, wh, wl) = the six MFMAs of tile i
template <int AHEAD, class Mma>
__device__ __forceinline__ void rl_stage(const unsigned char* st, Mma&& mma) {
    const int ln = opaque_lane();
    const int fo0 = stage_frag_off(ln), fo1 = fo0 ^ 64;
    constexpr int NB = AHEAD + 1;
    bf16x8 wh[NB], wl[NB];
    static_for<0, AHEAD>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        wh[n] = *reinterpret_cast<const bf16x8*>(st + n * 2048 + fo0); wl[n] = *reinterpret_cast<const bf16x8*>(st + n * 2048 + fo1);
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 8>([&](auto nc) {
        constexpr int n = decltype(nc)::value, nn = n + AHEAD;
        if constexpr (nn < 8 && (X3W_ABLATE & 128) == 0) {
            wh[nn % NB] = *reinterpret_cast<const bf16x8*>(st + nn * 2048 + fo0);
            wl[nn % NB] = *reinterpret_cast<const bf16x8*>(st + nn * 2048 + fo1);
        }
        mma(n, wh[n % NB], wl[n % NB]);
        if constexpr (nn < 8) {
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
// the wait in front of interval t's barrier: the pieces of intervals t + 1 .. may still be in flight (four per wave and interval)
__device__ __forceinline__ void rl_wait(int t, int T) {
    if (t + RL_P - 1 < T) wait_dma<4 * (RL_P - 1)>();
    else if (t + 1 < T) wait_dma<4>();
    else wait_dma<0>();
}

// acc: x of the wave's own 16 rows (in: after the attention branch; out: + fc2(gelu(fc1(LayerNorm2(x)))) + b2); ah / al: LayerNorm2 of the own rows.
// sb1 / sb2: the biases in LDS.  Every wave of the workgroup must call it; no LDS-DMA may be in flight on entry, none is on return.
template <int E, int AHEAD>
__device__ __forceinline__ void mlp_roles(unsigned char* ring, const float* sb1, const float* sb2, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          int w8, int lane, f32x4 (&acc)[E / 16], const bf16x8 (&ah)[E / 32], const bf16x8 (&al)[E / 32] X3W_TARG) {
    constexpr int F = 4 * E, NU = F / 32, NI = 3 * NU + 6;      // units; intervals (3 NU + 4, padded to a multiple of three)
    static_assert(E == 384 && NI % 3 == 0, "written for E = 384");
    const int i4 = w8 & 3, g = lane >> 4;
    const unsigned vk32 = voff_k32(lane, w8, E);
    const unsigned v128w = StreamLaneX::calc<2>(lane, w8 >> 1, E) + (unsigned)((w8 & 1) * 4 * 16 * E);
    unsigned char* hbuf = ring + RL_HBUF_OFF + i4 * 4096 + lane * 16;          // the pair's four fragments of h[u]: (row tile, hi | lo)
    // exchange buffers inside the ring's LDS (idle until the prefetch below): x tiles of half A | `a` fragments of half B, twelve KiB per wave and round
    unsigned char* xa = ring + i4 * 12288 + lane * 16;
    unsigned char* xb = ring + 49152 + i4 * 12288 + lane * 16;
    int role_a;      // through the scalar unit, so that the branch is a scalar branch: as a lane-mask condition the two role bodies get linearised (B's body, then a flag test, then A's
                     // body) and everything A needs stays live across B's loop — 96 registers too many
    asm volatile("s_cmp_lt_u32 %1, 4\n\ts_cselect_b32 %0, 1, 0" : "=s"(role_a) : "s"(w8) : "scc");
    if (role_a) {
        // ================================================================ half A: fc1 + GELU
        bf16x8 a2h[2][E / 32], a2l[2][E / 32];          // row tile 0: own rows, 1: the partner's
#pragma unroll
        for (int kb = 0; kb < E / 32; ++kb) { a2h[0][kb] = ah[kb]; a2l[0][kb] = al[kb]; }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int tl = 0; tl < 12; ++tl) *reinterpret_cast<f32x4*>(xa + tl * 1024) = acc[12 * r + tl];
            group_fence();
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                a2h[1][6 * r + k] = *reinterpret_cast<const bf16x8*>(xb + (2 * k) * 1024);
                a2l[1][6 * r + k] = *reinterpret_cast<const bf16x8*>(xb + (2 * k + 1) * 1024);
            }
            group_fence();
        }
        if (X3W_RL_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        static_for<0, RL_P>([&](auto tc) { rl_issue<E>(ring, wrsrc, w1_off, w2_off, vk32, v128w, w8, decltype(tc)::value); });
        f32x4 acc1[2][2];
        X3W_TICK(0);
        for (int j = 0; j < NI / 3; ++j) {
            static_for<0, 3>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                const int t = 3 * j + s;
                rl_wait(t, NI);
                group_fence();
                X3W_TICK(7);
                if constexpr (s == 0) {
                    if (j >= 1 && j <= NU) {          // GELU of unit j - 1 -> h fragments
                        const float* bp = sb1 + (j - 1) * 32 + 8 * g;
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            float v[8];
#pragma unroll
                            for (int q = 0; q < 4; ++q) { v[q] = x3w_gelu(acc1[0][rt][q] + bp[q]); v[4 + q] = x3w_gelu(acc1[1][rt][q] + bp[4 + q]); }
                            bf16x8 hh, hl;
                            split8(v, hh, hl);
                            *reinterpret_cast<bf16x8*>(hbuf + (2 * rt) * 1024) = hh;
                            *reinterpret_cast<bf16x8*>(hbuf + (2 * rt + 1) * 1024) = hl;
                        }
                    }
                    if (j < NU) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; PQ_X3W_ACC_PIN(acc1[i][0]); PQ_X3W_ACC_PIN(acc1[i][1]); }
                    }
                    X3W_TICK(8);
                }
                if (t + RL_P < NI) rl_issue<E>(ring, wrsrc, w1_off, w2_off, vk32, v128w, w8, t + RL_P);
                if (j < NU && (X3W_ABLATE & 512) == 0) {
                    rl_stage<AHEAD>(ring + (t & 3) * (2 * STAGE), [&](int i, const bf16x8& wh, const bf16x8& wl) {
                        const int kb = 4 * s + (i >> 1);
                        x3::mma3_w(acc1[i & 1][0], acc1[i & 1][1], wh, wl, a2h[0][kb], a2l[0][kb], a2h[1][kb], a2l[1][kb]);
                    });
                }
                X3W_TICK(9);
            });
        }
        if (X3W_RL_PRIO == 1) __builtin_amdgcn_s_setprio(X3W_PRIO ? 1 : 0);
        // x of the own rows comes back from the partner
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            group_fence();
#pragma unroll
            for (int tl = 0; tl < 12; ++tl) acc[12 * r + tl] = *reinterpret_cast<const f32x4*>(xa + tl * 1024);
            group_fence();
        }
        X3W_TICK(1);
    } else {
        // ================================================================ half B: fc2 into x
        f32x4 acc2[E / 16][2];                           // row tile 0: the partner's rows, 1: own rows
#pragma unroll
        for (int tl = 0; tl < E / 16; ++tl) acc2[tl][1] = acc[tl];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                *reinterpret_cast<bf16x8*>(xb + (2 * k) * 1024) = ah[6 * r + k];
                *reinterpret_cast<bf16x8*>(xb + (2 * k + 1) * 1024) = al[6 * r + k];
            }
            group_fence();
#pragma unroll
            for (int tl = 0; tl < 12; ++tl) acc2[12 * r + tl][0] = *reinterpret_cast<const f32x4*>(xa + tl * 1024);
            group_fence();
        }
        if (X3W_RL_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        static_for<0, RL_P>([&](auto tc) { rl_issue<E>(ring, wrsrc, w1_off, w2_off, vk32, v128w, w8, decltype(tc)::value); });
        bf16x8 hh[2], hl[2];
        X3W_TICK(0);
        for (int j = 0; j < NI / 3; ++j) {
            static_for<0, 3>([&](auto sc) {
                constexpr int s = decltype(sc)::value, sb = (s + 2) % 3;          // B's stage of its unit in this interval
                const int t = 3 * j + s;
                const int ub = s == 0 ? j - 2 : j - 1;                           // B's unit
                rl_wait(t, NI);
                group_fence();
                X3W_TICK(7);
                if constexpr (sb == 0) {
                    if (ub >= 0 && ub < NU) {
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            hh[rt] = *reinterpret_cast<const bf16x8*>(hbuf + (2 * rt) * 1024);
                            hl[rt] = *reinterpret_cast<const bf16x8*>(hbuf + (2 * rt + 1) * 1024);
                        }
                    }
                }
                if (t + RL_P < NI) rl_issue<E>(ring, wrsrc, w1_off, w2_off, vk32, v128w, w8, t + RL_P);
                if (ub >= 0 && ub < NU && (X3W_ABLATE & 1024) == 0) {
                    rl_stage<AHEAD>(ring + (t & 3) * (2 * STAGE) + STAGE, [&](int i, const bf16x8& wh, const bf16x8& wl) {
                        // hh / hl[0]: the half-A partner's rows (row tile 0 of the pair), [1]: own rows
                        x3::mma3_w(acc2[sb * 8 + i][0], acc2[sb * 8 + i][1], wh, wl, hh[0], hl[0], hh[1], hl[1]);
                    });
                }
                X3W_TICK(10);
            });
        }
        if (X3W_RL_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        // + b2, then x of the partner's rows goes back
        {
#pragma unroll
            for (int q32 = 0; q32 < E / 32; ++q32) {
                const float4 b0 = *reinterpret_cast<const float4*>(sb2 + 32 * q32 + 8 * g), b1 = *reinterpret_cast<const float4*>(sb2 + 32 * q32 + 8 * g + 4);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    f32x4& ta = acc2[(q32 >> 2) * 8 + 2 * (q32 & 3)][rt];
                    f32x4& tb = acc2[(q32 >> 2) * 8 + 2 * (q32 & 3) + 1][rt];
                    ta[0] += b0.x; ta[1] += b0.y; ta[2] += b0.z; ta[3] += b0.w;
                    tb[0] += b1.x; tb[1] += b1.y; tb[2] += b1.z; tb[3] += b1.w;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int tl = 0; tl < 12; ++tl) *reinterpret_cast<f32x4*>(xa + tl * 1024) = acc2[12 * r + tl][0];
            group_fence();
            group_fence();
        }
#pragma unroll
        for (int tl = 0; tl < E / 16; ++tl) acc[tl] = acc2[tl][1];
        X3W_TICK(1);
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
        issue_stage_k<1>(wrsrc, vpe, (hp.wpe + (unsigned)(ng * 128 * PK + kb * 32)) * 4u, 4u * PK, ring + st * STAGE + w8 * 2048, w8);
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
    issue_stage_k<0>(wrsrc, sl.template voff<0>(), (wkv_off + (unsigned)(c * 64 * E + (2 * pp + s) * 64)) * 4u, 4u * E, dst, w8);
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

// PHASES (diagnostic instantiations of tools/microbench/x3_variant.hip; the product is 3): bit 0 the attention branch, bit 1 the MLP branch
template <int E, int PHASES = 3>
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
    if (X3W_PRIO == 1 && w8 < 4) __builtin_amdgcn_s_setprio(1);
    if (X3W_PRIO == 2 && w8 >= 4) __builtin_amdgcn_s_setprio(1);      // the younger half (the arbitration loser at equal priority)
    if (X3W_PRIO == 3 && (w8 & 1)) __builtin_amdgcn_s_setprio(1);

    f32x4 acc[E / 16];
    bf16x8 ah[E / 32], al[E / 32];
#if X3W_TIMERS
    Timers x3t_storage; Timers* x3t = &x3t_storage;
#pragma unroll
    for (int i = 0; i < 13; ++i) x3t->acc[i] = 0;
    x3t->last = clock64();
#endif
    if (head.images) patch_head<E>(head, ring, wrsrc, w8, lane, blockIdx.x, acc);
    else load_x_to_acc<E>(x, m0, M, w8, rr, g, acc);
    float* xbuf = scratch + (size_t)blockIdx.x * (2 * 48 * 1024);      // the parked tiles of the residual stream (X3W_PARK_TILES x 512 lanes x 4 floats)
    float* obuf = xbuf + 48 * 1024;                                     // ... and the attention output fragments (24 pieces x 512 lanes x 4 floats)

    for (int l = 0; l < depth; ++l) {
        const EncBlockParams* bp = blocks + l;
        if constexpr ((PHASES & 1) != 0) {
            // ---- attention branch, head loop: parameters bqkv (3E) | bproj (E) | ln1 gamma (E) | ln1 beta (E)
            __syncthreads();
            heads_prefetch<E>(sl, ring, wrsrc, bp->wqkv, w8);
            params_to_lds(sph, pbase + bp->bqkv, 3 * E, tid);
            params_to_lds(sph + 3 * E, pbase + bp->bproj, E, tid);
            params_to_lds(sph + 4 * E, pbase + bp->ln1_w, E, tid);
            params_to_lds(sph + 5 * E, pbase + bp->ln1_b, E, tid);
            __syncthreads();
            ln_acc_to_frag<E>(acc, sph + 4 * E, sph + 5 * E, eps, g, ah, al);
            X3W_TICK(0);
            park_acc<E>(acc, xbuf, tid);
            X3W_TICK(1);
            heads_phase<E, X3W_AHEAD>(ring, img, sph, wrsrc, bp->wqkv, 0.125f, sl, w8, tid, ah, al, obuf X3W_TPASS);
            // ---- attention branch, proj: x and the O fragments come back (each lane re-reads what it wrote)
            __syncthreads();                                                // every wave is done with the K / V^T images and the ring
#if X3W_O_JIT
            // whatever the compiler itself moved out of the register file for the head loop comes back HERE, while no LDS-DMA is in flight: a scratch reload inside the proj
            // stream carries an s_waitcnt vmcnt(0) that drains the pieces issued just before it
#pragma unroll
            for (int i = X3W_PARK_TILES; i < E / 16; ++i) asm volatile("" : "+v"(acc[i]));
#endif
            proj_prefetch<E, RING>(sl, ring, wrsrc, bp->wproj, w8);
            // (the addresses go through an empty asm: the optimiser must not forward the stored values to these loads)
            const float* xback = xbuf; const float* oback = obuf;
            asm volatile("" : "+s"(xback), "+s"(oback) :: "memory");
            unpark_acc<E>(acc, xback, tid);
#pragma unroll
            for (int kb = 0; kb < (X3W_O_JIT ? 6 : E / 32); ++kb) {
                const float* o = oback + ((size_t)((kb * 2) * NT) + tid) * 4;
                ah[kb] = *reinterpret_cast<const bf16x8*>(o);
                al[kb] = *reinterpret_cast<const bf16x8*>(o + NT * 4);
            }
            X3W_TICK(1);
            proj_phase<E, RING, X3W_AHEAD>(ring, wrsrc, bp->wproj, sl, w8, ah, al, acc, oback, tid X3W_TPASS);
            add_bias_to_acc<E>(sph + 3 * E, g, acc);
            X3W_TICK(12);
        }
        if constexpr ((PHASES & 2) != 0) {
            // ---- MLP branch: parameters b1 (4E) | b2 (E) | ln2 gamma (E) | ln2 beta (E)
            __syncthreads();
            if constexpr (!X3W_MLP_ROLES) mlp_prefetch<E>(sl, ring, wrsrc, bp->w1, bp->w2, w8);
            params_to_lds(sp, pbase + bp->b1, F, tid);
            params_to_lds(sp + F, pbase + bp->b2, E, tid);
            params_to_lds(sp + F + E, pbase + bp->ln2_w, E, tid);
            params_to_lds(sp + F + 2 * E, pbase + bp->ln2_b, E, tid);
            __syncthreads();
            ln_acc_to_frag<E>(acc, sp + F + E, sp + F + 2 * E, eps, g, ah, al);
            X3W_TICK(0);
            if constexpr (X3W_MLP_ROLES) {
                mlp_roles<E, X3W_RL_AHEAD>(ring, sp, sp + F, wrsrc, bp->w1, bp->w2, w8, lane, acc, ah, al X3W_TPASS);
            } else {
                mlp_phase<E, X3W_AHEAD>(ring, sp, wrsrc, bp->w1, bp->w2, sl, w8, ah, al, acc X3W_TPASS);
                add_bias_to_acc<E>(sp + F, g, acc);
            }
            X3W_TICK(12);
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
    kv_phase<E, RING, X3W_AHEAD>(ring, sp, wrsrc, tail.wkv, sl, w8, blockIdx.x, tail.heads, tail.kmem, tail.vmem, tail.plane_elems, ah, al);
#if X3W_TIMERS
    X3W_TICK(12);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 13; ++i) reinterpret_cast<long long*>(x + ((size_t)m0 + 16 * w8) * E)[i] = x3t->acc[i];
    }
#endif
}

template <int E, int PHASES = 3>
hipError_t launch_enc_blocks_x3w(hipStream_t s, float* x, const void* wpack, size_t wbytes, const float* pbase, const EncBlockParams* blocks,
                                 int depth, float eps, int M, float* scratch, const EncTailX3& tail = EncTailX3{0, 0, 0, 0, nullptr, nullptr, 0},
                                 const EncHeadX3& head = EncHeadX3{nullptr, 0, 0, nullptr}) {
    constexpr size_t lds = x3::enc_blocks_x3_lds<E>();
    if (wbytes >= ((size_t)1 << 32) || M % 128 != 0) return hipErrorInvalidValue;
    auto kern = enc_blocks_x3w_kernel<E, PHASES>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(M / 128), dim3(NT), lds, s, x, reinterpret_cast<const unsigned char*>(wpack), (unsigned)wbytes, pbase, blocks, depth, eps, M, scratch, tail, head);
    return hipGetLastError();
}

}  // namespace x3w
}  // namespace pq
