// Fused AR decoder step (bf16 mode, one query per image): modules.py:55-98 query stream + decoder.norm + head
// (+ model.py:142-145 greedy pick) in TWO launches around the cross-attention instead of eight.
//
// Every decoder op except the cross-attention is local to one (image, query) row, so a workgroup that owns 16 rows (one
// MFMA M tile) carries them through a whole chain of GEMMs with the activations parked in LDS:
//
//   dec_step_pre_kernel   self-attention from the (position, token) tables -> out_proj + pos_queries residual ->
//                         norm1 -> cross-attention q-projection                                  writes t, qc
//   (dec_cross_attn_ar_kernel streams the image's K/V: the one part that needs the whole chip's HBM bandwidth)
//   dec_step_post_kernel  cross out_proj + residual -> norm2 -> linear1 + GELU -> linear2 + residual -> decoder.norm
//                         -> head -> arg-max / EOS bookkeeping                                   writes logits, tok
//
// At 512 rows an AR step is 32 workgroups: far too few to fill 256 CUs, but the step is a latency chain, not a throughput
// problem (1.9 GFLOP).  What the fusion removes is seven launch + drain + fill gaps per step (26 steps per batch), three
// dependent HBM round trips inside each small GEMM, and the round trips of t / xn / hdn between them.  The price is that
// every workgroup streams ALL decoder weights (0.6 MB pre, 3.0 MB post) through one CU, so the weight stream must run near
// the per-CU fill rate (tools/microbench/l2_fill.hip: 115-135 GB/s with global_load_lds rings, a third of that through
// register loads).  Each wave therefore owns a private ring of 1-KB LDS slots: `global_load_lds` copies the wave's next
// W fragments (lane l's 16 bytes land at slot + 16 l, exactly where lane l reads its MFMA operand back) while the wave
// multiplies — no block-level barrier inside a GEMM, only counted `s_waitcnt vmcnt`.
// Rounding points are the generic path's (decode_pass_e): bf16 GEMM operands, fp32 accumulation, fp32 residual stream.
#pragma once
#include "common.h"
#include "decoder_attn.h"
#include "encoder_mlp.h"        // static_for, wait_vmcnt

namespace pq {

constexpr int DS_ROWS = 16;     // rows per workgroup
constexpr int DS_NW = 8;        // waves per workgroup; 16-wide output column tiles are dealt round-robin to waves
#ifndef PQ_DS_RING
#define PQ_DS_RING 8
#endif
constexpr int DS_RING = PQ_DS_RING;      // 1-KB LDS slots per wave: fragment copies in flight (64 KB per CU)
constexpr int DS_LA = 2;        // fragments read ahead from LDS into registers (hides the ds_read latency behind MFMAs)

// DS_TIMERS=1 (diagnostic builds only, tools/ds_step_timers.py): wave 0 of workgroup 0 stamps s_memtime (100 MHz) after every phase of the two
// step kernels; the mlp kernel parks its stamps behind the partial sums, the next mid kernel writes both sets OVER row 0's logits of the
// position it finishes (slots 0-7 its own, 16-20 the mlp kernel's) — the picks come from LDS, so the decode itself is unchanged.
#ifndef DS_TIMERS
#define DS_TIMERS 0
#endif
#if DS_TIMERS
#define DS_T0() const unsigned long long ds_t0 = __builtin_amdgcn_s_memtime(); unsigned ds_tk[8] = {}
#define DS_TICK(i) ds_tk[i] = (unsigned)(__builtin_amdgcn_s_memtime() - ds_t0)
#else
#define DS_T0()
#define DS_TICK(i)
#endif

// Fragment-ordered weights.  A [N][K] row-major weight is re-packed once per weight set (frag_pack_kernel) into
//   Wp[tile = n / 16][kp = k / 64][half][lane = (n & 15) + 16 g][8]  =  W[16 tile + (lane & 15)][64 kp + 16 g + 8 half + 0..7]
// i.e. unit (tile, kp, half) is the 1-KB MFMA operand fragment of one wave, contiguous in memory, and the units of one
// tile follow each other: a wave's whole weight stream is one linear run of 1-KB `global_load_lds` copies.  (Read in
// place from the row-major weight the same fragment is 16 half-used cache lines: measured 4x slower end to end.)
// k is walked 64 at a time: lane group g takes k = 64 kp + 16 g + [0, 16) as the k-slots of TWO MFMAs (the slot
// assignment is free as long as both operands agree).  Rows past N are zero in the packed copy.
static __global__ __launch_bounds__(256)
void frag_pack_kernel(const bf16_t* __restrict__ W, int N, int K, int ldw, bf16_t* __restrict__ out, int tiles) {
    const int KP = K / 64;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece per thread
    if (idx >= (size_t)tiles * KP * 2 * 64) return;
    const int lane = (int)(idx & 63), half = (int)((idx >> 6) & 1);
    const int kp = (int)((idx >> 7) % KP), tile = (int)((idx >> 7) / KP);
    const int n = tile * 16 + (lane & 15), k = 64 * kp + 16 * (lane >> 4) + 8 * half;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N) v = *reinterpret_cast<const u32x4*>(W + (size_t)n * ldw + k);
    *reinterpret_cast<u32x4*>(out + idx * 8) = v;
}
constexpr size_t frag_pack_elems(int N, int K) { return (size_t)((N + 15) / 16) * 16 * K; }

// bf16x3 arithmetic: the same fragment order with every 1-KB unit followed by its residual plane,
//   Wp[tile][kp][half][plane][lane][8],  plane 0 = bf16(W), plane 1 = bf16(W - plane 0)
// (2 x frag_pack_elems bf16 elements), packed from the fp32 master.  A product is three MFMAs: W_hi A_lo + W_hi A_hi on the
// hi unit, W_lo A_hi on the lo unit that follows it in the stream; the activations are split where they are parked in LDS.
static __global__ __launch_bounds__(256)
void frag_pack_x3_kernel(const float* __restrict__ W, int N, int K, int ldw, bf16_t* __restrict__ out, int tiles) {
    const int KP = K / 64;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece per thread
    if (idx >= (size_t)tiles * KP * 4 * 64) return;
    const int lane = (int)(idx & 63), plane = (int)((idx >> 6) & 1), half = (int)((idx >> 7) & 1);
    const int kp = (int)((idx >> 8) % KP), tile = (int)((idx >> 8) / KP);
    const int n = tile * 16 + (lane & 15), k = 64 * kp + 16 * (lane >> 4) + 8 * half;
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float f = n < N ? W[(size_t)n * ldw + k + i] : 0.f;
        const bf16_t hi = static_cast<bf16_t>(f);
        o[i] = plane ? static_cast<bf16_t>(f - static_cast<float>(hi)) : hi;
    }
    *reinterpret_cast<bf16x8*>(out + idx * 8) = o;
}

// acc[t] += Wp[tile (w + DS_NW t)] (16 x K)  x  A^T (K x 16 rows).  Result layout: lane (m = lane & 15, g = lane >> 4)
// holds columns n = 16 tile + 4 g + r (r = 0..3) of row m.  Tiles past `tiles` recompute the last tile (discarded).
// Unit u = ((kp * TN) + t) * 2 + half is one 1-KB fragment; unit u lives in the wave's ring slot u % DS_RING.
// ds_prefetch<K, TN> issues the first DS_RING copies of a GEMM's weight stream; it may run as soon as the previous GEMM
// of the wave has consumed its last fragment (the weights do not depend on activations), i.e. before the epilogue,
// the block barrier and the LayerNorm that separate two GEMMs.  ds_wave_gemm<..., true> then skips its own prologue.
// kp0 / tile_kp: multiply against the k-slice [64 kp0, 64 kp0 + K) of a matrix packed with tile_kp k-chunks per tile
// (0 = the whole matrix, K / 64 chunks per tile).
// PL = planes per fragment: 1 (bf16 weights) or 2 (bf16x3: hi unit, lo unit); unit u = (((kp * TN) + t) * 2 + half) * PL + plane.
template <int K, int TN, int PL = 1>
__device__ __forceinline__ void ds_prefetch(const bf16_t* __restrict__ Wp, int tiles, int wave, unsigned char* wring, int kp0 = 0, int tile_kp = 0) {
    const int lane = threadIdx.x & 63;
    constexpr int KP = K / 64, U = KP * TN * 2 * PL, PRE = U < DS_RING ? U : DS_RING;
    if (tile_kp == 0) tile_kp = KP;
    static_for<0, PRE>([&](auto uc) {
        constexpr int u = decltype(uc)::value, kp = u / (2 * PL * TN), t = (u / (2 * PL)) % TN, half = (u / PL) & 1, plane = u % PL;
        int tile = wave + DS_NW * t;
        tile = tile < tiles ? tile : tiles - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wp + (size_t)tile * tile_kp * (1024 * PL) + lane * 8 + (((kp0 + kp) * 2 + half) * PL + plane) * 512),
                                         (__attribute__((address_space(3))) void*)(wring + (u % DS_RING) * 1024), 16, 0, 0);
    });
}

// PL == 2 (bf16x3): a_lds is the hi plane of the activations, a_lds + a_lo the lo plane (same pitch).
template <int K, int TN, bool PREFETCHED = false, int PL = 1>
__device__ __forceinline__ void ds_wave_gemm(const bf16_t* a_lds, int lda, const bf16_t* __restrict__ Wp, int tiles,
                                             int wave, unsigned char* wring, f32x4 (&acc)[TN], int kp0 = 0, int tile_kp = 0, int a_lo = 0) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    constexpr int KP = K / 64, U = KP * TN * 2 * PL, PRE = U < DS_RING ? U : DS_RING;
    if (tile_kp == 0) tile_kp = KP;
    const bf16_t* wp[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        int tile = wave + DS_NW * t;
        tile = tile < tiles ? tile : tiles - 1;
        wp[t] = Wp + (size_t)tile * tile_kp * (1024 * PL) + (size_t)kp0 * (1024 * PL) + lane * 8;
    }
    const bf16_t* ap = a_lds + r16 * lda + 16 * g;
    auto issue = [&](auto uc) {
        constexpr int u = decltype(uc)::value, kp = u / (2 * PL * TN), t = (u / (2 * PL)) % TN, half = (u / PL) & 1, plane = u % PL;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp[t] + ((kp * 2 + half) * PL + plane) * 512),
                                         (__attribute__((address_space(3))) void*)(wring + (u % DS_RING) * 1024), 16, 0, 0);
    };
    // Software pipeline per wave: unit u + DS_RING is copied global -> LDS while unit u + DS_LA is read LDS -> registers
    // and unit u is multiplied.  Before step u the copies of units < min(U, u + DS_RING) have been issued; reading unit v
    // needs all but the (issued - v - 1) newest of them landed.
    Frag<bf16_t> wbuf[DS_LA + 1];
    auto fetch = [&](auto vc, auto issued_c) {
        constexpr int v = decltype(vc)::value, issued = decltype(issued_c)::value;
        wait_vmcnt<issued - v - 1>();
        wbuf[v % (DS_LA + 1)].v = *reinterpret_cast<const bf16x8*>(wring + (v % DS_RING) * 1024 + lane * 16);
    };
    if constexpr (!PREFETCHED) static_for<0, PRE>(issue);
    static_for<0, (DS_LA < U ? DS_LA : U)>([&](auto vc) { fetch(vc, std::integral_constant<int, PRE>{}); });
    Frag<bf16_t> a0, a1, a0l, a1l;
    static_for<0, U>([&](auto uc) {
        constexpr int u = decltype(uc)::value, kp = u / (2 * PL * TN), t = (u / (2 * PL)) % TN, half = (u / PL) & 1, plane = u % PL;
        if constexpr (t == 0 && half == 0 && plane == 0) {
            a0.v = *reinterpret_cast<const bf16x8*>(ap + 64 * kp);
            a1.v = *reinterpret_cast<const bf16x8*>(ap + 64 * kp + 8);
            if constexpr (PL == 2) {
                a0l.v = *reinterpret_cast<const bf16x8*>(ap + a_lo + 64 * kp);
                a1l.v = *reinterpret_cast<const bf16x8*>(ap + a_lo + 64 * kp + 8);
            }
        }
        constexpr int issued = (u + DS_RING < U) ? u + DS_RING : U;
        if constexpr (u + DS_LA < U) fetch(std::integral_constant<int, u + DS_LA>{}, std::integral_constant<int, issued>{});
        if constexpr (PL == 2 && plane == 0) mma16(acc[t], wbuf[u % (DS_LA + 1)], half ? a1l : a0l);      // W_hi A_lo (small term first)
        mma16(acc[t], wbuf[u % (DS_LA + 1)], half ? a1 : a0);                                            // W_hi A_hi, or W_lo A_hi on a lo unit
        // slot u % DS_RING is free once unit u's fragment has been consumed by the MFMA above (order pinned below)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + DS_RING < U) issue(std::integral_constant<int, u + DS_RING>{});
    });
}

// LayerNorm of rows 2w, 2w + 1 of the fp32 residual tile (pitch PT) into the bf16 A buffer (pitch PA).
// X3: the row is written as a bf16 pair, hi plane at abuf, lo plane (value - hi) at abuf + a_lo.
template <int E, bool X3 = false>
__device__ __forceinline__ void ds_layernorm_rows(const float* tl, int PT, bf16_t* abuf, int PA, const float* __restrict__ gw,
                                                  const float* __restrict__ gb, float eps, int wave, int a_lo = 0) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int rr = 0; rr < DS_ROWS / DS_NW; ++rr) {
        const int row = wave * (DS_ROWS / DS_NW) + rr;
        float v[E / 64], s = 0.f;
#pragma unroll
        for (int i = 0; i < E / 64; ++i) { v[i] = tl[row * PT + lane + 64 * i]; s += v[i]; }
        const float mean = wave_sum(s) * (1.0f / E);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < E / 64; ++i) { const float d = v[i] - mean; ss += d * d; }
        const float var = wave_sum(ss) * (1.0f / E) + eps;
        const float rstd = X3 ? 1.0f / sqrtf(var) : __builtin_amdgcn_rsqf(var);
#pragma unroll
        for (int i = 0; i < E / 64; ++i) {
            const int c = lane + 64 * i;
            const float y = (v[i] - mean) * rstd * gw[c] + gb[c];
            const bf16_t hi = from_f32<bf16_t>(y);
            abuf[row * PA + c] = hi;
            if constexpr (X3) abuf[a_lo + row * PA + c] = from_f32<bf16_t>(y - to_f32(hi));
        }
    }
}

// Table self-attention of R rows by one wave, registers only (decoder_attn.h: dec_self_attn_kernel): lane l works for head
// h = l >> 2: it scores keys j = (l & 3) + 4 c, the quad reduces max / sum over DPP, and the same lane then mixes value
// columns d = 8 l .. 8 l + 7 (which belong to head l >> 2) with probabilities quad-broadcast.  tokv[r]: lane j holds token j
// of row r's context (j < Lk).  Result: bf16 row of E values at arow + r * lda (LDS).
// X3: the K | V table is f32 and the result row is written as a bf16 pair (hi at arow, lo at arow + a_lo).
// The gathers are the phase's whole cost (a third of the mid kernel when every key's value row was its own round trip behind its own
// branch: tools/ds_step_timers.py), so they go out in groups: the R rows' score lookups together, then the value rows of KG keys of all
// R rows per uniform branch, the first group before the scores are even looked up.  Keys past Lk in a group read key Lk - 1's row
// against a probability of exactly zero, so every accumulator sees the same fmaf chain as one key at a time (acc + 0 * v == acc).
#ifndef PQ_DS_SA_KG
#define PQ_DS_SA_KG 0           // keys per group of value gathers; 0 = 8 (f32 table) / 16 (bf16 table): 64 value registers per row in flight
#endif
template <int E, bool X3 = false, int R = 1>
__device__ __forceinline__ void ds_self_attn_rows(const float* __restrict__ stab, const typename std::conditional<X3, float, bf16_t>::type* __restrict__ kvtab,
                                                  const int (&tokv)[R], int ntok, int npos, int Lk, int pos, bf16_t* arow, int lda, int a_lo = 0) {
    static_assert(DEC_HD == 32 && DEC_MAXL == 32, "lane mapping assumes 32-wide heads and <= 32 keys");
    constexpr int H = E / DEC_HD;
    constexpr int KG = PQ_DS_SA_KG ? PQ_DS_SA_KG : (X3 ? 8 : 16);
    const int lane = threadIdx.x & 63;
    const int h = lane >> 2, q = lane & 3;
    const bool live = h < H;
    const int d0 = live ? 8 * lane : 0;
    using VT = typename std::conditional<X3, f32x4, bf16x8>::type;
    constexpr int VP = X3 ? 2 : 1;                // 16-byte pieces per lane and key
    VT v[R][KG][VP];
    auto gather = [&](int g0) {                   // value rows of keys g0 .. g0 + KG - 1 of every row (clamped to the last key)
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const int jc = min(g0 + k, Lk - 1);
                const int tj = __builtin_amdgcn_readlane(tokv[r], jc);
                const VT* vp = reinterpret_cast<const VT*>(kvtab + (unsigned)((jc * ntok + tj) * (2 * E) + E) + d0);      // 32-bit on purpose: the table is npos x ntok x 2E elements
#pragma unroll
                for (int i = 0; i < VP; ++i) v[r][k][i] = vp[i];
            }
        }
    };
    gather(0);                                    // they depend on the tokens only: in flight under the score lookups and the soft-max
    float sc[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int j = q + 4 * c;
            const int tj = __shfl(tokv[r], j, 64);
            sc[r][c] = (live && j < Lk) ? stab[(unsigned)(((pos * npos + j) * ntok + tj) * H + h)] : -INFINITY;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) mx = fmaxf(mx, sc[r][c]);
        mx = fmaxf(mx, dpp_mov<0xB1>(mx));
        mx = fmaxf(mx, dpp_mov<0x4E>(mx));
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { sc[r][c] = (q + 4 * c < Lk) ? expf(sc[r][c] - mx) : 0.f; sum += sc[r][c]; }
        sum += dpp_mov<0xB1>(sum);
        sum += dpp_mov<0x4E>(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int c = 0; c < 8; ++c) sc[r][c] *= inv;
    }
    float acc[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r][i] = 0.f;
    auto mix = [&](auto g0c) {
        constexpr int g0 = decltype(g0c)::value;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                // key j = 4 c + qq: its probability sits in lane (quad base + qq), register sc[c] (zero past Lk)
                const int c = (g0 + k) >> 2, qq = (g0 + k) & 3;
                const float pj = qq == 0 ? dpp_mov<0x00>(sc[r][c]) : qq == 1 ? dpp_mov<0x55>(sc[r][c]) : qq == 2 ? dpp_mov<0xAA>(sc[r][c]) : dpp_mov<0xFF>(sc[r][c]);
                if constexpr (X3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc[r][i] = fmaf(pj, v[r][k][0][i], acc[r][i]); acc[r][4 + i] = fmaf(pj, v[r][k][1][i], acc[r][4 + i]); }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[r][i] = fmaf(pj, to_f32(v[r][k][0][i]), acc[r][i]);
                }
            }
        }
    };
    mix(std::integral_constant<int, 0>{});
    static_for<1, 32 / KG>([&](auto gc) {
        constexpr int g0 = decltype(gc)::value * KG;
        if (g0 < Lk) {
            gather(g0);
            mix(std::integral_constant<int, g0>{});
        }
    });
    if (live) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bf16x8 o, ol;
#pragma unroll
            for (int i = 0; i < 8; ++i) { o[i] = from_f32<bf16_t>(acc[r][i]); ol[i] = from_f32<bf16_t>(acc[r][i] - to_f32(o[i])); }
            *reinterpret_cast<bf16x8*>(arow + r * lda + d0) = o;
            if constexpr (X3) *reinterpret_cast<bf16x8*>(arow + r * lda + a_lo + d0) = ol;
        }
    }
}

template <int E> constexpr size_t dec_step_pre_lds() { return (size_t)DS_ROWS * ((E + 8) * 2 + (E + 4) * 4) + (size_t)DS_NW * DS_RING * 1024; }
template <int E> constexpr size_t dec_step_post_lds() {
    return (size_t)DS_ROWS * ((E + 8) * 2 + (4 * E + 8) * 2 + (E + 4) * 4) + (size_t)DS_NW * DS_RING * 1024;
}

// ---- before the cross-attention ---------------------------------------------------------------------------------------
// tok [M][ldt]; keys j < Lk are the content tokens; query position `pos` for every row (AR step: Lq == 1, no masks).
// Wo, Wq: fragment-packed (Wq = first E rows of cross_attn.in_proj_weight; bq its first E biases).  t_out, qc_out: fp32 [M][E].
template <int E>
__global__ __launch_bounds__(64 * DS_NW)
void dec_step_pre_kernel(const float* __restrict__ stab, const bf16_t* __restrict__ kvtab, const int* __restrict__ tok, int ldt,
                         int ntok, int npos, int Lk, int pos, const bf16_t* __restrict__ Wo, const float* __restrict__ bo,
                         const float* __restrict__ posq, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                         const bf16_t* __restrict__ Wq, const float* __restrict__ bq, float* __restrict__ t_out,
                         float* __restrict__ qc_out, int M) {
    constexpr int H = E / DEC_HD, PA = E + 8, PT = E + 4, TILES = E / 16, TN = (TILES + DS_NW - 1) / DS_NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ds[];
    bf16_t* abuf = reinterpret_cast<bf16_t*>(smem_ds);                 // [DS_ROWS][PA]
    float* tl = reinterpret_cast<float*>(abuf + DS_ROWS * PA);         // [DS_ROWS][PT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    unsigned char* wring = reinterpret_cast<unsigned char*>(tl + DS_ROWS * PT) + wave * DS_RING * 1024;
    const int row0 = blockIdx.x * DS_ROWS;

    // self-attention of rows 2w, 2w + 1, one wave per row
    ds_prefetch<E, TN>(Wo, TILES, wave, wring);
    {
        constexpr int RPW = DS_ROWS / DS_NW;
        int tokv[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int b = min(row0 + wave * RPW + rr, M - 1);
            tokv[rr] = lane < Lk ? tok[(size_t)b * ldt + lane] : 0;
        }
        ds_self_attn_rows<E, false, RPW>(stab, kvtab, tokv, ntok, npos, Lk, pos, abuf + wave * RPW * PA, PA);
    }
    __syncthreads();

    // t = pos_queries[pos] + sa @ Wo^T + bo
    {
        f32x4 acc[TN] = {};
        ds_wave_gemm<E, TN, true>(abuf, PA, Wo, TILES, wave, wring, acc);
        ds_prefetch<E, TN>(Wq, TILES, wave, wring);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES) {
                const int n = tile * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(bo + n);
                const float4 pv = *reinterpret_cast<const float4*>(posq + n);
                f32x4 o = {acc[t][0] + bv.x + pv.x, acc[t][1] + bv.y + pv.y, acc[t][2] + bv.z + pv.z, acc[t][3] + bv.w + pv.w};
                *reinterpret_cast<f32x4*>(tl + r16 * PT + n) = o;
                if (row0 + r16 < M) *reinterpret_cast<f32x4*>(t_out + (size_t)(row0 + r16) * E + n) = o;
            }
        }
    }
    __syncthreads();
    ds_layernorm_rows<E>(tl, PT, abuf, PA, ln_w, ln_b, eps, wave);
    __syncthreads();
    // qc = norm1(t) @ Wq^T + bq
    {
        f32x4 acc[TN] = {};
        ds_wave_gemm<E, TN, true>(abuf, PA, Wq, TILES, wave, wring, acc);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES && row0 + r16 < M) {
                const int n = tile * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(bq + n);
                f32x4 o = {acc[t][0] + bv.x, acc[t][1] + bv.y, acc[t][2] + bv.z, acc[t][3] + bv.w};
                *reinterpret_cast<f32x4*>(qc_out + (size_t)(row0 + r16) * E + n) = o;
            }
        }
    }
}

// ---- after the cross-attention ----------------------------------------------------------------------------------------
// ca bf16 [M][E]; t fp32 [M][E] (from the pre kernel); Wco, W1, W2, Wh fragment-packed; logits fp32 [M][Ltot][C] at position `pos`.
// argmax_mode: 0 = none, 1 = write tok[b][pos + 1], 2 = also keep the batch-level EOS bookkeeping (rowops.h: ar_argmax_kernel).
template <int E>
__global__ __launch_bounds__(64 * DS_NW)
void dec_step_post_kernel(const bf16_t* __restrict__ ca, const float* __restrict__ t_in, const bf16_t* __restrict__ Wco,
                          const float* __restrict__ bco, const float* __restrict__ ln2_w, const float* __restrict__ ln2_b,
                          const bf16_t* __restrict__ W1, const float* __restrict__ b1, const bf16_t* __restrict__ W2,
                          const float* __restrict__ b2, const float* __restrict__ lnf_w, const float* __restrict__ lnf_b, float eps,
                          const bf16_t* __restrict__ Wh, const float* __restrict__ bh, int C, float* __restrict__ logits, int Ltot,
                          int pos, int M, int argmax_mode, int* __restrict__ tok, int ldt, int eos_id,
                          unsigned char* __restrict__ eos_seen, int* __restrict__ eos_rows, int* __restrict__ ar_len) {
    constexpr int F = 4 * E, PA = E + 8, PH = F + 8, PT = E + 4, TILES = E / 16, TN = (TILES + DS_NW - 1) / DS_NW;
    constexpr int TN1 = F / 16 / DS_NW;                       // linear1: column tiles per wave
    constexpr int PL = 128;                                  // logits tile pitch (C <= 128); aliases hbuf after linear2
    static_assert(F % (16 * DS_NW) == 0, "linear1 width must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ds[];
    bf16_t* abuf = reinterpret_cast<bf16_t*>(smem_ds);                 // [DS_ROWS][PA]
    bf16_t* hbuf = abuf + DS_ROWS * PA;                                // [DS_ROWS][PH]
    float* tl = reinterpret_cast<float*>(hbuf + DS_ROWS * PH);         // [DS_ROWS][PT]
    float* lg = reinterpret_cast<float*>(hbuf);                        // [DS_ROWS][PL]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    unsigned char* wring = reinterpret_cast<unsigned char*>(tl + DS_ROWS * PT) + wave * DS_RING * 1024;
    const int row0 = blockIdx.x * DS_ROWS;

    ds_prefetch<E, TN>(Wco, TILES, wave, wring);
    // stage the 16 rows of ca (bf16) and t (fp32)
    for (int i = threadIdx.x; i < DS_ROWS * (E / 8); i += 64 * DS_NW) {
        const int row = i / (E / 8), c = (i - row * (E / 8)) * 8;
        const int gr = min(row0 + row, M - 1);
        *reinterpret_cast<bf16x8*>(abuf + row * PA + c) = *reinterpret_cast<const bf16x8*>(ca + (size_t)gr * E + c);
    }
    for (int i = threadIdx.x; i < DS_ROWS * (E / 4); i += 64 * DS_NW) {
        const int row = i / (E / 4), c = (i - row * (E / 4)) * 4;
        const int gr = min(row0 + row, M - 1);
        *reinterpret_cast<f32x4*>(tl + row * PT + c) = *reinterpret_cast<const f32x4*>(t_in + (size_t)gr * E + c);
    }
    __syncthreads();
    // t += ca @ Wco^T + bco
    {
        f32x4 acc[TN] = {};
        ds_wave_gemm<E, TN, true>(abuf, PA, Wco, TILES, wave, wring, acc);
        ds_prefetch<E, TN1>(W1, F / 16, wave, wring);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES) {
                const int n = tile * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(bco + n);
                f32x4* p = reinterpret_cast<f32x4*>(tl + r16 * PT + n);
                f32x4 o = *p;
                o[0] += acc[t][0] + bv.x; o[1] += acc[t][1] + bv.y; o[2] += acc[t][2] + bv.z; o[3] += acc[t][3] + bv.w;
                *p = o;
            }
        }
    }
    __syncthreads();
    ds_layernorm_rows<E>(tl, PT, abuf, PA, ln2_w, ln2_b, eps, wave);
    __syncthreads();
    // h = gelu(norm2(t) @ W1^T + b1)
    {
        f32x4 acc[TN1] = {};
        ds_wave_gemm<E, TN1, true>(abuf, PA, W1, F / 16, wave, wring, acc);
        ds_prefetch<F, TN>(W2, TILES, wave, wring);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int n = (wave + DS_NW * t) * 16 + 4 * g;
            const float4 bv = *reinterpret_cast<const float4*>(b1 + n);
            const float o[4] = {gelu_poly(acc[t][0] + bv.x), gelu_poly(acc[t][1] + bv.y), gelu_poly(acc[t][2] + bv.z), gelu_poly(acc[t][3] + bv.w)};
            store4<bf16_t>(hbuf + r16 * PH + n, o);
        }
    }
    __syncthreads();
    // t += h @ W2^T + b2
    {
        f32x4 acc[TN] = {};
        ds_wave_gemm<F, TN, true>(hbuf, PH, W2, TILES, wave, wring, acc);
        ds_prefetch<E, 1>(Wh, (C + 15) / 16, wave, wring);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES) {
                const int n = tile * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(b2 + n);
                f32x4* p = reinterpret_cast<f32x4*>(tl + r16 * PT + n);
                f32x4 o = *p;
                o[0] += acc[t][0] + bv.x; o[1] += acc[t][1] + bv.y; o[2] += acc[t][2] + bv.z; o[3] += acc[t][3] + bv.w;
                *p = o;
            }
        }
    }
    __syncthreads();
    ds_layernorm_rows<E>(tl, PT, abuf, PA, lnf_w, lnf_b, eps, wave);
    __syncthreads();
    // logits = decoder.norm(t) @ Wh^T + bh   (C <= 128 classes: one column tile per wave); hbuf is free: lg aliases it
    {
        f32x4 acc[1] = {};
        ds_wave_gemm<E, 1, true>(abuf, PA, Wh, (C + 15) / 16, wave, wring, acc);
        const int n = wave * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r < C) {
                const float v = acc[0][r] + bh[n + r];
                lg[r16 * PL + n + r] = v;
                if (row0 + r16 < M) logits[((size_t)(row0 + r16) * Ltot + pos) * C + n + r] = v;
            }
        }
    }
    if (argmax_mode == 0) return;
    __syncthreads();
    // greedy pick (first maximum, as torch.argmax) + batch-level EOS bookkeeping
#pragma unroll 1
    for (int rr = 0; rr < DS_ROWS / DS_NW; ++rr) {
        const int row = wave * (DS_ROWS / DS_NW) + rr, b = row0 + row;
        if (b >= M) continue;
        float best = -INFINITY; int bi = ARGMAX_NONE;
        for (int c = lane; c < C; c += 64) {
            const float v = lg[row * PL + c];
            if (argmax_take(v, c, best, bi)) { best = v; bi = c; }
        }
        wave_argmax(best, bi);
        bi = argmax_final(bi, C);
        if (lane == 0) {
            tok[(size_t)b * ldt + pos + 1] = bi;
            if (argmax_mode == 2 && bi == eos_id && !eos_seen[b]) {
                eos_seen[b] = 1;
                const int done = atomicAdd(eos_rows, 1) + 1;
                if (done == M) *ar_len = pos + 1;
            }
        }
    }
}

// =====================================================================================================================
// AR loop, second arrangement (forward_impl): the step boundary moves to the cross-attention.
//
//   dec_step_mid_kernel   finishes step i-1 (linear2 partial sums + residual -> decoder.norm -> head -> logits, greedy pick,
//                         EOS bookkeeping) AND starts step i (self-attention with the token it has just picked ->
//                         out_proj -> norm1 -> q-projection), 16 rows per workgroup
//   dec_cross_attn_ar_kernel (step i)
//   dec_step_mlp_kernel   cross out_proj + residual -> norm2 -> linear1 + GELU -> linear2 for a SIXTH of the hidden units per
//                         workgroup (ds_split<E>() workgroups per row tile; a quarter until round 4): each streams 0.3 + 2 x 0.1 MB of
//                         weights (bf16; twice that as bf16 pairs) instead of 3 MB; the partial products of linear2 go to global memory and are summed, in a fixed order, by
//                         the next mid kernel
// Same three launches per step as the pre / post arrangement, but the longest weight stream per CU drops from 3.0 MB to
// 0.7 MB and the token never leaves the workgroup between the pick and the next self-attention.
// =====================================================================================================================
#ifndef PQ_DS_SPLIT_384
#define PQ_DS_SPLIT_384 6       // 4 before round 4's timers: every split streams the whole cross out_proj, so the shares of linear1 / linear2 are what shrinks
#endif
template <int E> constexpr int ds_split() { return E >= 384 ? PQ_DS_SPLIT_384 : 2; }   // workgroups per row tile in dec_step_mlp_kernel

// X3 (bf16x3 arithmetic): every bf16 activation buffer is a hi plane followed by a lo plane
// The small per-column parameters (LayerNorm affines, biases, the position query) are staged into LDS by the first instructions of the
// launch — one round trip shared with the activation loads — instead of being fetched where they are used: each such use was its own
// exposed L2 round trip in the middle of the chain (tools/ds_step_timers.py: ~1 us per LayerNorm for 0.3 us of arithmetic).
template <int E> constexpr int ds_mid_params() { return 7 * E + 128; }      // lnf w | lnf b | head bias (128) | bo | pos query | ln1 w | ln1 b | bq
template <int E> constexpr int ds_mlp_params() { return 3 * E + 4 * E / ds_split<E>(); }   // bco | ln2 w | ln2 b | this split's b1
template <int E, bool X3 = false> constexpr size_t dec_step_mid_lds() {
    return (size_t)DS_ROWS * ((E + 8) * 2 * (X3 ? 2 : 1) + (E + 4) * 4 + 128 * 4) + (size_t)ds_mid_params<E>() * 4 + (size_t)DS_NW * DS_RING * 1024;
}
template <int E, bool X3 = false> constexpr size_t dec_step_mlp_lds() {
    return (size_t)DS_ROWS * (((E + 8) * 2 + (4 * E / ds_split<E>() + 8) * 2) * (X3 ? 2 : 1) + (E + 4) * 4) + (size_t)ds_mlp_params<E>() * 4 + (size_t)DS_NW * DS_RING * 1024;
}
// dst[k][0, n[k]) = src[k][0, valid[k]) followed by zeros, n[k] <= threads per workgroup.  Two halves: the loads go out first thing in the
// launch, branch-free (clamped index), so all segments share one round trip with the activation loads that follow them; the LDS stores
// come after those have been issued.
template <int NSEG>
__device__ __forceinline__ void ds_params_load(float (&v)[NSEG], const float* const (&src)[NSEG], const int (&valid)[NSEG]) {
#pragma unroll
    for (int k = 0; k < NSEG; ++k) v[k] = src[k][min((int)threadIdx.x, valid[k] - 1)];
}
template <int NSEG>
__device__ __forceinline__ void ds_params_store(const float (&v)[NSEG], float* const (&dst)[NSEG], const int (&n)[NSEG], const int (&valid)[NSEG]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NSEG; ++k)
        if (t < n[k]) dst[k][t] = t < valid[k] ? v[k] : 0.f;
}

// ---- the mid kernel's start half split over DS_QS workgroups per row tile (round 4) ---------------------------------------------------------
// out_proj and the q-projection are two 0.3 MB (bf16; 0.6 MB as pairs) weight streams through ONE CU each, 10 of the mid kernel's 25 us.
// With QS > 1, workgroup qs of a row tile owns columns [qs E/QS, (qs + 1) E/QS) of x = pos_query + sa @ Wo^T + bo: it streams only those
// rows of Wo, and only those K-columns of Wq — the q-projection becomes a K-split whose partial sums the cross-attention kernel adds up
// (decoder_attn.h QAsm / q_issue / q_finish; the launch boundary is the exchange, no grid-wide hand-off inside a launch).  norm1 sits between the
// two products and needs the row's mean and variance: the mean is known BEFORE x is, mean(x) = c0[pos] + sa . wbar with wbar the column
// means of Wo (dec_qfold_kernel, folded once per weight set), so every workgroup centres its own columns, multiplies them by ln_w, and
// leaves sum (x - m) and sum (x - m)^2 of its columns for the consumer, which finishes LayerNorm behind the product (two-pass arithmetic on
// centred values; the estimate's rounding error d is corrected exactly).  The finish half (partial sums -> decoder.norm -> head -> pick)
// and the table self-attention are computed by every workgroup of the tile (each needs the picked token and the whole sa row); only
// workgroup 0 writes logits / tok / EOS bookkeeping.
// (DS_QS = 3, decoder_attn.h: E / DS_QS must be a multiple of 64, the k-chunks of the fragment packs — 128 columns at E = 384, 64 at E = 192.)

// qfold = [wbar E | cq E | bq2 E | c0 npos]:  wbar[k] = mean_n Wo[n][k] on the operand values the product sees (T = bf16: the rounded
// weights), cq[n] = sum_k ln_w[k] Wq[n][k], bq2[n] = sum_k ln_b[k] Wq[n][k] + bq[n], c0[pos] = mean_n (pos_queries[pos][n] + bo[n]).
template <typename T>
__global__ __launch_bounds__(512)
void dec_qfold_kernel(const T* __restrict__ Wo, const float* __restrict__ Wq, const float* __restrict__ bq, const float* __restrict__ ln_w,
                      const float* __restrict__ ln_b, const float* __restrict__ bo, const float* __restrict__ pos_queries, int E, int npos,
                      float* __restrict__ qfold) {
    const int t = threadIdx.x;
    if (t < E) {
        float sw = 0.f, sc = 0.f, sb = 0.f;
        for (int n = 0; n < E; ++n) sw += to_f32(Wo[(size_t)n * E + t]);
        for (int k = 0; k < E; ++k) { const float w = Wq[(size_t)t * E + k]; sc = fmaf(ln_w[k], w, sc); sb = fmaf(ln_b[k], w, sb); }
        qfold[t] = sw / (float)E; qfold[E + t] = sc; qfold[2 * E + t] = sb + bq[t];
    }
    if (t < npos) {
        float s0 = 0.f;
        for (int n = 0; n < E; ++n) s0 += pos_queries[(size_t)t * E + n] + bo[n];
        qfold[3 * E + t] = s0 / (float)E;
    }
}

// tq: fp32 [M][E], t' of the step being finished (written by dec_step_mlp_kernel split 0); partial: fp32 [DS_SPLIT][M][E].
// pos: the step being started (its query position); the step being finished is pos - 1.  Lk = pos + 1 context tokens.
// X3: bf16x3 arithmetic — Wh / Wo / Wq are frag_pack_x3_kernel packs, kvtab is f32, every MFMA operand a bf16 pair; the
// element-wise parts (table soft-max, LayerNorm, residuals, pick) are the same fp32 code.
template <int E, bool X3 = false, int QS = 1>
__global__ __launch_bounds__(64 * DS_NW)
void dec_step_mid_kernel(int do_finish, int do_start, int pos, int M,
                         // finish
                         const float* __restrict__ tq, const float* __restrict__ partial, const float* __restrict__ b2,
                         const float* __restrict__ lnf_w, const float* __restrict__ lnf_b, float eps, const bf16_t* __restrict__ Wh,
                         const float* __restrict__ bh, int C, float* __restrict__ logits, int Ltot, int argmax_mode, int eos_id,
                         unsigned char* __restrict__ eos_seen, int* __restrict__ eos_rows, int* __restrict__ ar_len,
                         // start
                         const float* __restrict__ stab, const typename std::conditional<X3, float, bf16_t>::type* __restrict__ kvtab, int* __restrict__ tok, int ldt, int ntok,
                         int npos, const bf16_t* __restrict__ Wo, const float* __restrict__ bo, const float* __restrict__ pos_queries,
                         const float* __restrict__ ln1_w, const float* __restrict__ ln1_b, const bf16_t* __restrict__ Wq,
                         const float* __restrict__ bq, float* __restrict__ t_out, float* __restrict__ qc_out,
                         const float* __restrict__ qfold, float* __restrict__ qstats) {
    constexpr int PA = E + 8, PT = E + 4, TILES = E / 16, TN = (TILES + DS_NW - 1) / DS_NW, PL = 128, RPW = DS_ROWS / DS_NW;
    constexpr int DS_SPLIT = ds_split<E>();
    constexpr int EN = E / QS, TS = EN / 16, TNS = (TS + DS_NW - 1) / DS_NW;      // QS > 1: this workgroup's columns of x, their tiles
    static_assert(QS == 1 || EN % 64 == 0, "a split's columns are whole k-chunks of the fragment packs");
    constexpr int NPL = X3 ? 2 : 1, ALO = DS_ROWS * PA;                 // planes per operand; element offset of the lo plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ds[];
    bf16_t* abuf = reinterpret_cast<bf16_t*>(smem_ds);                 // [NPL][DS_ROWS][PA]
    float* tl = reinterpret_cast<float*>(abuf + NPL * DS_ROWS * PA);    // [DS_ROWS][PT]
    float* lg = tl + DS_ROWS * PT;                                     // [DS_ROWS][PL]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    float* prm = lg + DS_ROWS * PL;                                    // [ds_mid_params<E>()]
    float* const s_lnf_w = prm, * const s_lnf_b = prm + E, * const s_bh = prm + 2 * E, * const s_bo = prm + 2 * E + 128, * const s_posq = prm + 3 * E + 128,
         * const s_ln1_w = prm + 4 * E + 128, * const s_ln1_b = prm + 5 * E + 128, * const s_bq = prm + 6 * E + 128;
    unsigned char* wring = reinterpret_cast<unsigned char*>(prm + ds_mid_params<E>()) + wave * DS_RING * 1024;
    const int qs = QS > 1 ? (int)(blockIdx.x % QS) : 0;                // QS > 1: grid = row tiles x QS, this workgroup's share of the start half
    const int row0 = (int)(blockIdx.x / QS) * DS_ROWS;
    if (QS > 1 && !do_start && qs != 0) return;                        // the trailing finish-only launch is workgroup 0's alone
    int picked[RPW], tokv[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        picked[rr] = -1;
        // the context tokens of the step being started are in memory already, all but the one this launch is about to pick: their
        // load goes out first, one round trip fewer in front of the self-attention
        tokv[rr] = (do_start && lane <= pos) ? tok[(size_t)min(row0 + wave * RPW + rr, M - 1) * ldt + lane] : 0;
    }
    DS_T0();
    static_assert(E <= 64 * DS_NW, "one parameter per thread and segment");
    float* const prm_dst[8] = {s_lnf_w, s_lnf_b, s_bh, s_bo, s_posq, s_ln1_w, s_ln1_b, s_bq};
    // QS > 1: norm1's shift and the q-projection's bias are folded into the consumer's bq2; the slot of ln1_b holds wbar
    const float* const prm_src[8] = {lnf_w, lnf_b, bh, bo, pos_queries + (size_t)min(pos, npos - 1) * E, ln1_w, QS > 1 ? qfold : ln1_b, bq};
    const float c0 = QS > 1 ? qfold[3 * E + min(pos, npos - 1)] : 0.f;
    const int prm_n[8] = {E, E, 128, E, E, E, E, E}, prm_valid[8] = {E, E, C, E, E, E, E, E};
    float prm_v[8];
    ds_params_load<8>(prm_v, prm_src, prm_valid);

    if (do_finish) {
        // t'' = t' + b2 + partial[0] + partial[1] + ...   (fixed order: deterministic).  These loads go out BEFORE the head's
        // weight prefetch: vector memory returns in order, and a load queued behind eight LDS-DMA copies waits for all of them.
        for (int i = threadIdx.x; i < DS_ROWS * (E / 4); i += 64 * DS_NW) {
            const int row = i / (E / 4), c = (i - row * (E / 4)) * 4;
            const size_t gr = (size_t)min(row0 + row, M - 1) * E + c;
            f32x4 v = *reinterpret_cast<const f32x4*>(tq + gr);
            const float4 bv = *reinterpret_cast<const float4*>(b2 + c);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
#pragma unroll
            for (int sp = 0; sp < DS_SPLIT; ++sp) v += *reinterpret_cast<const f32x4*>(partial + (size_t)sp * M * E + gr);
            *reinterpret_cast<f32x4*>(tl + row * PT + c) = v;
        }
        ds_params_store<8>(prm_v, prm_dst, prm_n, prm_valid);
        ds_prefetch<E, 1, NPL>(Wh, (C + 15) / 16, wave, wring);
        __syncthreads();
        DS_TICK(0);
        ds_layernorm_rows<E, X3>(tl, PT, abuf, PA, s_lnf_w, s_lnf_b, eps, wave, ALO);
        __syncthreads();
        DS_TICK(1);
        {
            f32x4 acc[1] = {};
            ds_wave_gemm<E, 1, true, NPL>(abuf, PA, Wh, (C + 15) / 16, wave, wring, acc, 0, 0, ALO);
            if (do_start) {
                if constexpr (QS > 1) ds_prefetch<E, TNS, NPL>(Wo + (size_t)qs * TS * (E / 64) * (1024 * NPL), TS, wave, wring);
                else ds_prefetch<E, TN, NPL>(Wo, TILES, wave, wring);
            }
            const int n = wave * 16 + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < C) {
                    const float v = acc[0][r] + s_bh[n + r];
                    lg[r16 * PL + n + r] = v;
                    if (qs == 0 && row0 + r16 < M) logits[((size_t)(row0 + r16) * Ltot + (pos - 1)) * C + n + r] = v;
                }
            }
        }
        __syncthreads();                                     // lg complete; abuf free for the next step's self-attention
        DS_TICK(2);
        if (argmax_mode) {
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int row = wave * RPW + rr, b = row0 + row;
                float best = -INFINITY; int bi = ARGMAX_NONE;
                for (int c = lane; c < C; c += 64) {
                    const float v = lg[row * PL + c];
                    if (argmax_take(v, c, best, bi)) { best = v; bi = c; }
                }
                wave_argmax(best, bi);
                bi = argmax_final(bi, C);
                picked[rr] = bi;
                if (qs == 0 && lane == 0 && b < M) {
                    tok[(size_t)b * ldt + pos] = bi;
                    if (argmax_mode == 2 && bi == eos_id && !eos_seen[b]) {
                        eos_seen[b] = 1;
                        const int done = atomicAdd(eos_rows, 1) + 1;
                        if (done == M) *ar_len = pos;       // = (finished step) + 1
                    }
                }
            }
        }
    } else if (do_start) {
        ds_params_store<8>(prm_v, prm_dst, prm_n, prm_valid);
        if constexpr (QS > 1) ds_prefetch<E, TNS, NPL>(Wo + (size_t)qs * TS * (E / 64) * (1024 * NPL), TS, wave, wring);
        else ds_prefetch<E, TN, NPL>(Wo, TILES, wave, wring);
    }
    if (!do_start) return;
    DS_TICK(3);

    const int Lk = pos + 1;
    // self-attention of rows 2w, 2w + 1: context tokens from the load at the top, the newest one straight from the pick above
    {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
            if (picked[rr] >= 0 && lane == pos) tokv[rr] = picked[rr];
        ds_self_attn_rows<E, X3, RPW>(stab, kvtab, tokv, ntok, npos, Lk, pos, abuf + wave * RPW * PA, PA, ALO);
    }
    __syncthreads();
    DS_TICK(4);
    if constexpr (QS > 1) {
        const int n0 = qs * EN;
        float* mt = lg;                                      // [DS_ROWS] row-mean estimates (the logits tile is dead after the pick)
        // m = c0[pos] + sa . wbar  (= the mean of the row of x this tile's workgroups are about to compute, to rounding)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = wave * RPW + rr;
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < E / 64; ++i) {
                const int c = lane + 64 * i;
                float a = to_f32(abuf[row * PA + c]);
                if constexpr (X3) a += to_f32(abuf[ALO + row * PA + c]);
                dot = fmaf(a, s_ln1_b[c], dot);              // the slot holds wbar
            }
            dot = wave_sum(dot);
            if (lane == 0) mt[row] = dot + c0;
        }
        {   // x[:, cols] = pos_queries[pos] + sa @ Wo[cols]^T + bo
            f32x4 acc[TNS] = {};
            ds_wave_gemm<E, TNS, true, NPL>(abuf, PA, Wo + (size_t)qs * TS * (E / 64) * (1024 * NPL), TS, wave, wring, acc, 0, 0, ALO);
            ds_prefetch<EN, TN, NPL>(Wq, TILES, wave, wring, qs * (EN / 64), E / 64);
#pragma unroll
            for (int t = 0; t < TNS; ++t) {
                const int tile = wave + DS_NW * t;
                if (tile < TS) {
                    const int nl = tile * 16 + 4 * g, n = n0 + nl;
                    const float4 bv = *reinterpret_cast<const float4*>(s_bo + n);
                    const float4 pv = *reinterpret_cast<const float4*>(s_posq + n);
                    f32x4 o = {acc[t][0] + bv.x + pv.x, acc[t][1] + bv.y + pv.y, acc[t][2] + bv.z + pv.z, acc[t][3] + bv.w + pv.w};
                    *reinterpret_cast<f32x4*>(tl + r16 * PT + nl) = o;
                    if (row0 + r16 < M) *reinterpret_cast<f32x4*>(t_out + (size_t)(row0 + r16) * E + n) = o;
                }
            }
        }
        __syncthreads();
        DS_TICK(5);
        // centred, scaled columns as the q-projection's A operand (local columns 0 .. EN - 1); the column sums for the consumer
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = wave * RPW + rr;
            const float m = mt[row];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < EN / 64; ++i) {
                const int c = lane + 64 * i;
                const float d = tl[row * PT + c] - m;
                s1 += d; s2 = fmaf(d, d, s2);
                const float y = d * s_ln1_w[n0 + c];
                const bf16_t hi = from_f32<bf16_t>(y);
                abuf[row * PA + c] = hi;
                if constexpr (X3) abuf[ALO + row * PA + c] = from_f32<bf16_t>(y - to_f32(hi));
            }
            s1 = wave_sum(s1); s2 = wave_sum(s2);
            if (lane == 0 && row0 + row < M) *reinterpret_cast<float2*>(qstats + ((size_t)qs * M + row0 + row) * 2) = make_float2(s1, s2);
        }
        __syncthreads();
        DS_TICK(6);
        {   // qp[qs] = ((x - m) * ln_w)[:, cols] @ Wq[:, cols]^T
            f32x4 acc[TN] = {};
            ds_wave_gemm<EN, TN, true, NPL>(abuf, PA, Wq, TILES, wave, wring, acc, qs * (EN / 64), E / 64, ALO);
            float* dst = qc_out + (size_t)qs * M * E;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int tile = wave + DS_NW * t;
                if (tile < TILES && row0 + r16 < M) *reinterpret_cast<f32x4*>(dst + (size_t)(row0 + r16) * E + tile * 16 + 4 * g) = acc[t];
            }
        }
    } else {
    {   // t = pos_queries[pos] + sa @ Wo^T + bo
        const float* posq = s_posq;
        f32x4 acc[TN] = {};
        ds_wave_gemm<E, TN, true, NPL>(abuf, PA, Wo, TILES, wave, wring, acc, 0, 0, ALO);
        ds_prefetch<E, TN, NPL>(Wq, TILES, wave, wring);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES) {
                const int n = tile * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(s_bo + n);
                const float4 pv = *reinterpret_cast<const float4*>(posq + n);
                f32x4 o = {acc[t][0] + bv.x + pv.x, acc[t][1] + bv.y + pv.y, acc[t][2] + bv.z + pv.z, acc[t][3] + bv.w + pv.w};
                *reinterpret_cast<f32x4*>(tl + r16 * PT + n) = o;
                if (row0 + r16 < M) *reinterpret_cast<f32x4*>(t_out + (size_t)(row0 + r16) * E + n) = o;
            }
        }
    }
    __syncthreads();
    DS_TICK(5);
    ds_layernorm_rows<E, X3>(tl, PT, abuf, PA, s_ln1_w, s_ln1_b, eps, wave, ALO);
    __syncthreads();
    DS_TICK(6);
    {   // qc = norm1(t) @ Wq^T + bq
        f32x4 acc[TN] = {};
        ds_wave_gemm<E, TN, true, NPL>(abuf, PA, Wq, TILES, wave, wring, acc, 0, 0, ALO);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES && row0 + r16 < M) {
                const int n = tile * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(s_bq + n);
                f32x4 o = {acc[t][0] + bv.x, acc[t][1] + bv.y, acc[t][2] + bv.z, acc[t][3] + bv.w};
                *reinterpret_cast<f32x4*>(qc_out + (size_t)(row0 + r16) * E + n) = o;
            }
        }
    }
    }
#if DS_TIMERS
    DS_TICK(7);
    if (do_finish && blockIdx.x == 0 && threadIdx.x == 0) {
        float* dst = logits + (size_t)(pos - 1) * C;
        for (int i = 0; i < 8; ++i) dst[i] = (float)ds_tk[i];
        for (int i = 0; i < 5; ++i) dst[16 + i] = partial[(size_t)DS_SPLIT * M * E + i];
    }
#endif
}

// grid: row tiles x DS_SPLIT.  ca bf16 [M][E]; t_in fp32 [M][E] (from the mid kernel); tq_out fp32 [M][E] receives
// t' = t + ca @ Wco^T + bco (split 0 only); partial fp32 [DS_SPLIT][M][E] receives this split's share of h @ W2^T.
// X3: ca is f32 (the f32 cross-attention kernel's output), split into a bf16 pair as it is staged; GELU is the fp32 mode's
// erf form; Wco / W1 / W2 are frag_pack_x3_kernel packs.
template <int E, bool X3 = false>
__global__ __launch_bounds__(64 * DS_NW)
void dec_step_mlp_kernel(const typename std::conditional<X3, float, bf16_t>::type* __restrict__ ca, const float* __restrict__ t_in, const bf16_t* __restrict__ Wco,
                         const float* __restrict__ bco, const float* __restrict__ ln2_w, const float* __restrict__ ln2_b, float eps,
                         const bf16_t* __restrict__ W1, const float* __restrict__ b1, const bf16_t* __restrict__ W2,
                         float* __restrict__ tq_out, float* __restrict__ partial, int M) {
    constexpr int DS_SPLIT = ds_split<E>();
    constexpr int F = 4 * E, FS = F / DS_SPLIT, PA = E + 8, PH = FS + 8, PT = E + 4, TILES = E / 16, TN = (TILES + DS_NW - 1) / DS_NW;
    constexpr int TN1 = FS / 16 / DS_NW;                      // linear1: column tiles per wave in this split
    static_assert(FS % (16 * DS_NW) == 0 && FS % 64 == 0, "hidden width must split evenly over the workgroups and waves");
    constexpr int PL = X3 ? 2 : 1, ALO = DS_ROWS * PA, HLO = DS_ROWS * PH;   // planes per operand; element offsets of the lo planes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ds[];
    bf16_t* abuf = reinterpret_cast<bf16_t*>(smem_ds);                 // [PL][DS_ROWS][PA]
    bf16_t* hbuf = abuf + PL * DS_ROWS * PA;                           // [PL][DS_ROWS][PH]
    float* tl = reinterpret_cast<float*>(hbuf + PL * DS_ROWS * PH);    // [DS_ROWS][PT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    float* prm = tl + DS_ROWS * PT;                                    // [ds_mlp_params<E>()]
    float* const s_bco = prm, * const s_ln2_w = prm + E, * const s_ln2_b = prm + 2 * E, * const s_b1 = prm + 3 * E;
    unsigned char* wring = reinterpret_cast<unsigned char*>(prm + ds_mlp_params<E>()) + wave * DS_RING * 1024;
    const int rt = blockIdx.x / DS_SPLIT, sp = blockIdx.x - rt * DS_SPLIT;
    const int row0 = rt * DS_ROWS;
    DS_T0();
    static_assert(E <= 64 * DS_NW && FS <= 64 * DS_NW, "one parameter per thread and segment");
    float* const prm_dst[4] = {s_bco, s_ln2_w, s_ln2_b, s_b1};
    const float* const prm_src[4] = {bco, ln2_w, ln2_b, b1 + sp * FS};
    const int prm_n[4] = {E, E, E, FS};
    float prm_v[4];
    ds_params_load<4>(prm_v, prm_src, prm_n);

    for (int i = threadIdx.x; i < DS_ROWS * (E / 8); i += 64 * DS_NW) {
        const int row = i / (E / 8), c = (i - row * (E / 8)) * 8;
        const int gr = min(row0 + row, M - 1);
        if constexpr (X3) {
            const f32x4* src = reinterpret_cast<const f32x4*>(ca + (size_t)gr * E + c);
            const f32x4 v0 = src[0], v1 = src[1];
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = from_f32<bf16_t>(v0[j]); lo[j] = from_f32<bf16_t>(v0[j] - to_f32(hi[j]));
                hi[4 + j] = from_f32<bf16_t>(v1[j]); lo[4 + j] = from_f32<bf16_t>(v1[j] - to_f32(hi[4 + j]));
            }
            *reinterpret_cast<bf16x8*>(abuf + row * PA + c) = hi;
            *reinterpret_cast<bf16x8*>(abuf + ALO + row * PA + c) = lo;
        } else {
            *reinterpret_cast<bf16x8*>(abuf + row * PA + c) = *reinterpret_cast<const bf16x8*>(ca + (size_t)gr * E + c);
        }
    }
    for (int i = threadIdx.x; i < DS_ROWS * (E / 4); i += 64 * DS_NW) {
        const int row = i / (E / 4), c = (i - row * (E / 4)) * 4;
        const int gr = min(row0 + row, M - 1);
        *reinterpret_cast<f32x4*>(tl + row * PT + c) = *reinterpret_cast<const f32x4*>(t_in + (size_t)gr * E + c);
    }
    ds_params_store<4>(prm_v, prm_dst, prm_n, prm_n);
    ds_prefetch<E, TN, PL>(Wco, TILES, wave, wring);           // after the activation loads (in-order return, see the mid kernel)
    __syncthreads();
    DS_TICK(0);
    {   // t' = t + ca @ Wco^T + bco   (every split needs it for norm2; split 0 publishes it)
        f32x4 acc[TN] = {};
        ds_wave_gemm<E, TN, true, PL>(abuf, PA, Wco, TILES, wave, wring, acc, 0, 0, ALO);
        ds_prefetch<E, TN1, PL>(W1 + (size_t)sp * (FS / 16) * (E / 64) * (1024 * PL), FS / 16, wave, wring);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES) {
                const int n = tile * 16 + 4 * g;
                const float4 bv = *reinterpret_cast<const float4*>(s_bco + n);
                f32x4* p = reinterpret_cast<f32x4*>(tl + r16 * PT + n);
                f32x4 o = *p;
                o[0] += acc[t][0] + bv.x; o[1] += acc[t][1] + bv.y; o[2] += acc[t][2] + bv.z; o[3] += acc[t][3] + bv.w;
                *p = o;
                if (sp == 0 && row0 + r16 < M) *reinterpret_cast<f32x4*>(tq_out + (size_t)(row0 + r16) * E + n) = o;
            }
        }
    }
    __syncthreads();
    DS_TICK(1);
    ds_layernorm_rows<E, X3>(tl, PT, abuf, PA, s_ln2_w, s_ln2_b, eps, wave, ALO);
    __syncthreads();
    DS_TICK(2);
    {   // h[:, split] = gelu(norm2(t') @ W1[split]^T + b1[split])
        f32x4 acc[TN1] = {};
        ds_wave_gemm<E, TN1, true, PL>(abuf, PA, W1 + (size_t)sp * (FS / 16) * (E / 64) * (1024 * PL), FS / 16, wave, wring, acc, 0, 0, ALO);
        ds_prefetch<FS, TN, PL>(W2, TILES, wave, wring, sp * (FS / 64), F / 64);
#pragma unroll
        for (int t = 0; t < TN1; ++t) {
            const int n = (wave + DS_NW * t) * 16 + 4 * g;
            const float4 bv = *reinterpret_cast<const float4*>(s_b1 + n);
            if constexpr (X3) {
                const float o[4] = {gelu_erf(acc[t][0] + bv.x), gelu_erf(acc[t][1] + bv.y), gelu_erf(acc[t][2] + bv.z), gelu_erf(acc[t][3] + bv.w)};
                float ol[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ol[j] = o[j] - to_f32(from_f32<bf16_t>(o[j]));
                store4<bf16_t>(hbuf + r16 * PH + n, o);
                store4<bf16_t>(hbuf + HLO + r16 * PH + n, ol);
            } else {
                const float o[4] = {gelu_poly(acc[t][0] + bv.x), gelu_poly(acc[t][1] + bv.y), gelu_poly(acc[t][2] + bv.z), gelu_poly(acc[t][3] + bv.w)};
                store4<bf16_t>(hbuf + r16 * PH + n, o);
            }
        }
    }
    __syncthreads();
    DS_TICK(3);
    {   // partial[split] = h[:, split] @ W2[:, split]^T
        f32x4 acc[TN] = {};
        ds_wave_gemm<FS, TN, true, PL>(hbuf, PH, W2, TILES, wave, wring, acc, sp * (FS / 64), F / 64, HLO);
        float* dst = partial + (size_t)sp * M * E;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int tile = wave + DS_NW * t;
            if (tile < TILES && row0 + r16 < M) {
                const int n = tile * 16 + 4 * g;
                *reinterpret_cast<f32x4*>(dst + (size_t)(row0 + r16) * E + n) = acc[t];
            }
        }
    }
#if DS_TIMERS
    DS_TICK(4);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 5; ++i) partial[(size_t)DS_SPLIT * M * E + i] = (float)ds_tk[i];
#endif
}

}  // namespace pq
