// Decoder attention kernels (nn.MultiheadAttention semantics), restructured for the
// depth-1 two-stream decoder.
//
// Self-attention.  The content stream of a 1-layer decoder is a pure function of (position, token id), and the
// position queries are batch-independent, so everything except the soft-max and the value mix is tabulated once per
// weight set:
//     kvtab[pos][tok][2E]   K | V projection of norm_c(content(pos, tok))                  (T)
//     qself[pos][E]         (Wq norm_q(pos_queries[pos]) + bq) / sqrt(hd)                   (fp32)
//     stab[i][j][tok][h]    qself[i, head h] . K(kvtab[j][tok], head h)                     (fp32)
// so an AR step's self-attention is: gather <= 26 x H scores, soft-max, mix <= 26 gathered V rows.
//
// Cross-attention.  K/V of `memory` are projected ONCE per image by the encoder tail (the reference re-projects them
// on each of its 26 + refine_iters decoder calls) and stored head-split:  kmem[b][h][key][32], vmem[b][h][key][32].
//   * AR step (one query per image): dec_cross_attn_ar_kernel — memory-bound streaming of the image's 2 x 96 KB of K/V
//     as contiguous 1-KB wave loads, one workgroup of E threads per image.
//   * refinement / NAR (all <= 32 positions at once): dec_cross_attn_multi_kernel — one workgroup per (image, head), K
//     and V^T staged in LDS as fp32, scores and the value mix as register-blocked FMA loops (2.6 GFLOP per 512 images:
//     not worth an MFMA pipeline).
#pragma once
#include "common.h"

namespace pq {

constexpr int DEC_MAXL = 32;       // max context length / query count (LDT pitch of the token arrays)
constexpr int DEC_HD = 32;         // decoder head dim (E / dec_heads) — 384/12 and 192/6

// stab[((i * npos + j) * ntok + tok) * H + h]
template <typename T>
__global__ __launch_bounds__(256)
void score_table_kernel(const float* __restrict__ qself, const T* __restrict__ kvtab, float* __restrict__ stab,
                        int npos, int ntok, int E) {
    const int H = E / DEC_HD;
    const size_t total = (size_t)npos * npos * ntok * H;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int h = (int)(idx % H);
    size_t r = idx / H;
    const int tok = (int)(r % ntok); r /= ntok;
    const int j = (int)(r % npos);
    const int i = (int)(r / npos);
    const float* q = qself + (size_t)i * E + h * DEC_HD;
    const T* k = kvtab + ((size_t)j * ntok + tok) * (2 * E) + h * DEC_HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DEC_HD; ++d) s = fmaf(q[d], to_f32(k[d]), s);
    stab[idx] = s;
}

// One workgroup of E threads per (image b, query index qi); query position pos = i0 + qi; keys j < Lk are the content
// tokens tok[b][j].  Masks follow torch: qmask[pos * ldq + j] != 0 or kpm[b * ldk + j] != 0  =>  key j gets -inf.
template <typename T, int E>
__global__ __launch_bounds__(E)
void dec_self_attn_kernel(const float* __restrict__ stab, const T* __restrict__ kvtab, const int* __restrict__ tok,
                          int ldt, int ntok, int npos, const unsigned char* __restrict__ qmask, int ldq,
                          const unsigned char* __restrict__ kpm, int ldk, int Lk, int i0, int Lq, T* __restrict__ out,
                          const float* __restrict__ uq = nullptr) {
    // uq != nullptr: caller-supplied queries (model.decode's tgt_query, model.py:100-102) — uq[w][E] is their q-projection
    // (norm_q, in_proj rows 0..E, bias, 1/sqrt(hd) applied) and the scores are dot products against the content-key table
    // instead of look-ups in the position-query score table.
    constexpr int H = E / DEC_HD;
    __shared__ int stok[DEC_MAXL];
    __shared__ float sp[H][DEC_MAXL];
    const int t = threadIdx.x;
    const int w = blockIdx.x;
    const int b = w / Lq, qi = w - b * Lq, pos = i0 + qi;
    if (t < Lk) stok[t] = tok[(size_t)b * ldt + t];
    __syncthreads();
    if (t < H * Lk) {
        const int j = t / H, h = t - j * H;
        const bool masked = (qmask && qmask[(size_t)pos * ldq + j]) || (kpm && kpm[(size_t)b * ldk + j]);
        float sc;
        if (uq) {
            const float* qv = uq + (size_t)w * E + h * DEC_HD;
            const T* kv = kvtab + ((size_t)j * ntok + stok[j]) * (2 * E) + h * DEC_HD;
            sc = 0.f;
#pragma unroll
            for (int d = 0; d < DEC_HD; ++d) sc = fmaf(qv[d], to_f32(kv[d]), sc);
        } else {
            sc = stab[(((size_t)pos * npos + j) * ntok + stok[j]) * H + h];
        }
        sp[h][j] = masked ? -INFINITY : sc;
    }
    __syncthreads();
    if (t < H) {                              // 12 x 26 values: a serial soft-max per head is a few hundred cycles
        float mx = -INFINITY;
        for (int j = 0; j < Lk; ++j) mx = fmaxf(mx, sp[t][j]);
        float sum = 0.f;
        for (int j = 0; j < Lk; ++j) { const float p = expf(sp[t][j] - mx); sp[t][j] = p; sum += p; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < Lk; ++j) sp[t][j] *= inv;
    }
    __syncthreads();
    const int h = t / DEC_HD;
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j)
        acc = fmaf(sp[h][j], to_f32(kvtab[((size_t)j * ntok + stok[j]) * (2 * E) + E + t]), acc);
    out[(size_t)w * E + t] = from_f32<T>(acc);
}

// Wave-per-row variant of dec_self_attn_kernel for <= 16 heads: no LDS, no block barriers.
// Lane l works for head h = l >> 2: it scores keys j = (l & 3) + 4 c (c < 8), the quad reduces max / sum over DPP, and the
// same lane then mixes value columns d = 8 l .. 8 l + 7 (which belong to head l >> 2), probabilities quad-broadcast.
// T = bf16_t (bf16 mode) or float (bf16x3 mode: f32 tables and output, the same arithmetic as dec_self_attn_kernel<float>).
template <int E, typename T = bf16_t>
__global__ __launch_bounds__(256)
void dec_self_attn_wave_kernel(const float* __restrict__ stab, const T* __restrict__ kvtab, const int* __restrict__ tok,
                               int ldt, int ntok, int npos, const unsigned char* __restrict__ qmask, int ldq,
                               const unsigned char* __restrict__ kpm, int ldk, int Lk, int i0, int Lq, T* __restrict__ out,
                               int rows) {
    constexpr int H = E / DEC_HD;
    static_assert(H <= 16 && DEC_HD == 32 && DEC_MAXL == 32, "lane mapping: <= 16 heads of 32, <= 32 keys");
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows) return;
    const int b = w / Lq, qi = w - b * Lq, pos = i0 + qi;
    const int tokv = lane < Lk ? tok[(size_t)b * ldt + lane] : 0;
    const int h = lane >> 2, q = lane & 3;
    const bool live = h < H;
    float sc[8], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int j = q + 4 * c;
        const int tj = __shfl(tokv, j, 64);
        bool ok = live && j < Lk;
        if (ok && qmask && qmask[(size_t)pos * ldq + j]) ok = false;
        if (ok && kpm && kpm[(size_t)b * ldk + j]) ok = false;
        sc[c] = ok ? stab[(((size_t)pos * npos + j) * ntok + tj) * H + h] : -INFINITY;
        mx = fmaxf(mx, sc[c]);
    }
    mx = fmaxf(mx, dpp_mov<0xB1>(mx));
    mx = fmaxf(mx, dpp_mov<0x4E>(mx));
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc[c] = (sc[c] == -INFINITY) ? 0.f : expf(sc[c] - mx); sum += sc[c]; }
    sum += dpp_mov<0xB1>(sum);
    sum += dpp_mov<0x4E>(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < 8; ++c) sc[c] *= inv;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int d0 = live ? 8 * lane : 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float pq4[4] = {dpp_mov<0x00>(sc[c]), dpp_mov<0x55>(sc[c]), dpp_mov<0xAA>(sc[c]), dpp_mov<0xFF>(sc[c])};
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int j = 4 * c + qq;
            if (j < Lk) {
                const int tj = __builtin_amdgcn_readlane(tokv, j);
                if constexpr (sizeof(T) == 4) {
                    const f32x4* vp = reinterpret_cast<const f32x4*>(kvtab + ((size_t)j * ntok + tj) * (2 * E) + E + d0);
                    const f32x4 v0 = vp[0], v1 = vp[1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc[i] = fmaf(pq4[qq], v0[i], acc[i]); acc[4 + i] = fmaf(pq4[qq], v1[i], acc[4 + i]); }
                } else {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(kvtab + ((size_t)j * ntok + tj) * (2 * E) + E + d0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(pq4[qq], to_f32(v[i]), acc[i]);
                }
            }
        }
    }
    if (live) {
        if constexpr (sizeof(T) == 4) {
            f32x4* op = reinterpret_cast<f32x4*>(out + (size_t)w * E + d0);
            op[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
            op[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
        } else {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = from_f32<bf16_t>(acc[i]);
            *reinterpret_cast<bf16x8*>(out + (size_t)w * E + d0) = o;
        }
    }
}

// AR-step cross-attention: one workgroup of E threads per image, ONE query.  qc fp32 [B][E] un-scaled projected query;
// kmem, vmem T [B][H][Nk][32] (head-split, key rows of 32 contiguous d); out T [B][E].  Nk = 128 memory tokens.
// Pure streaming of the image's K and V (196 KB in bf16): every wave-level load is one contiguous 1 KB — lane
// (kl = lane / LPR, dl = lane % LPR) takes the 16-byte piece dl of key row (16 c + kl) — and all 16 loads of a head are
// issued before the first score is needed.  A wave owns a head at a time: partial dot products are reduced over the LPR
// lanes of a key row with DPP, the soft-max over the 128 keys and the value mix over the key lanes with DPP rotations
// inside a 16-lane row and two cross-row shuffles.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) { return v + dpp_mov<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ float dpp_max(float v) { return fmaxf(v, dpp_mov<CTRL>(v)); }

// Where an AR step's cross-attention query comes from.  nsplit == 0: qc [M][E] as the q-projection wrote it.  nsplit > 0 (decoder_step.h,
// dec_step_mid_kernel<.., QS > 1>): the step's out_proj -> norm1 -> q-projection chain was split over nsplit workgroups per row tile, each
// owning E / nsplit columns of x = out_proj(...) + residual; what they left is, per split s, qp[s][M][E] = ((x - m) * ln_w)[:, cols of s] @
// Wq[:, cols of s]^T with m an estimate of the row mean good to rounding, and stats[s][M][2] = sum (x - m), sum (x - m)^2 over the
// split's columns.  With d = sum_s S1 / E (the estimate's error), var = sum_s S2 / E - d^2:
//     q = rsqrt(var + eps) * (sum_s qp[s] - d * cq) + bq2,     cq = Wq @ ln_w,   bq2 = Wq @ ln_b + bq   (folded once per weight set)
// — LayerNorm's two-pass arithmetic on centred values, with the division by the standard deviation moved behind the product.
struct QAsm {
    const float* qp = nullptr; const float* stats = nullptr; const float* cq = nullptr; const float* bq2 = nullptr;
    int nsplit = 0, M = 0; float inv_e = 0.f, eps = 0.f;
};
constexpr int DS_QS = 3;        // workgroups per row tile of the split mid kernel (decoder_step.h); E / DS_QS must be a multiple of 64
// How the two AR kernels read it: thread t of the image's workgroup loads what element t of the query needs as the launch's FIRST loads
// (one dword per split: coalesced, not one 16-byte piece per lane and head — that form cost the kernel 2-3 us of load issue), every wave then
// puts its first head's K / V loads in flight, and only then is the element finished, parked in LDS and the workgroup synchronised.  Vector
// memory returns in order: query loads issued behind the K / V stream, or one round trip per split inside a loop, wait for the whole
// stream first (measured: + 5.7 us per step); a run-time branch on the form between the loads and their use made the scheduler sink the
// K / V loads into the arithmetic (half the bytes in flight) — hence one instantiation per form and a sched_barrier behind the loads.
template <bool SPLIT>
struct QOne { float part[SPLIT ? DS_QS : 1]; float2 st[DS_QS]; float cq, bq; };
template <bool SPLIT>
__device__ __forceinline__ void q_issue(const QAsm& qa, const float* __restrict__ qc, int b, int E, int n, QOne<SPLIT>& r) {
    if constexpr (!SPLIT) {
        r.part[0] = qc[(size_t)b * E + n];
    } else {
#pragma unroll
        for (int s = 0; s < DS_QS; ++s) {
            r.st[s] = *reinterpret_cast<const float2*>(qa.stats + ((size_t)s * qa.M + b) * 2);
            r.part[s] = qa.qp[((size_t)s * qa.M + b) * E + n];
        }
        r.cq = qa.cq[n]; r.bq = qa.bq2[n];
    }
}
template <bool SPLIT>
__device__ __forceinline__ float q_finish(const QAsm& qa, const QOne<SPLIT>& r, float scale) {
    if constexpr (!SPLIT) {
        return r.part[0] * scale;
    } else {
        float s1 = 0.f, s2 = 0.f, q = r.part[0];
#pragma unroll
        for (int s = 0; s < DS_QS; ++s) { s1 += r.st[s].x; s2 += r.st[s].y; }
#pragma unroll
        for (int s = 1; s < DS_QS; ++s) q += r.part[s];
        const float d = s1 * qa.inv_e;
        const float rstd = 1.0f / sqrtf(s2 * qa.inv_e - d * d + qa.eps);
        return (rstd * (q - d * r.cq) + r.bq) * scale;
    }
}

template <typename T, int E, bool QSPLIT = false>
__global__ __launch_bounds__(E)
void dec_cross_attn_ar_kernel(const float* __restrict__ qc, const QAsm qa, const T* __restrict__ kmem, const T* __restrict__ vmem,
                              float scale, T* __restrict__ out) {
    constexpr int H = E / DEC_HD, NK = 128;
    constexpr int EPC = 16 / (int)sizeof(T);           // elements per 16-byte piece: 8 (bf16) or 4 (f32)
    constexpr int LPR = DEC_HD / EPC;                  // lanes per key row: 4 or 8
    constexpr int KPL = 64 / LPR;                      // key rows per wave-level load: 16 or 8
    constexpr int NL = NK / KPL;                       // loads per head and operand: 8 or 16
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = E / 64, b = blockIdx.x;
    const int kl = lane / LPR, dl = lane % LPR;
    // sum / max over the lanes that hold the same d piece (all kl): rotations by multiples of LPR inside the 16-lane row,
    // then the other three rows
    auto over_keys_sum = [](float v) {
        if constexpr (LPR == 4) v = dpp_add<0x124>(v);     // row_ror:4
        v = dpp_add<0x128>(v);                             // row_ror:8
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
    auto over_keys_max = [](float v) {
        if constexpr (LPR == 4) v = dpp_max<0x124>(v);
        v = dpp_max<0x128>(v);
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        return v;
    };
    __shared__ __attribute__((aligned(16))) float qs[E];
    QOne<QSPLIT> q1;
    q_issue<QSPLIT>(qa, qc, b, E, (int)threadIdx.x, q1);
    bool first = true;
    for (int h = wid; h < H; h += nw) {              // exactly two heads per wave (H = 2 nw): the barrier below is uniform
        const T* kb = kmem + (((size_t)b * H + h) * NK + kl) * DEC_HD + dl * EPC;
        const T* vb = vmem + (((size_t)b * H + h) * NK + kl) * DEC_HD + dl * EPC;
        union Piece { u32x4 u; T e[EPC]; };
        Piece kr[NL], vr[NL];
#pragma unroll
        // streamed once per step and 100 MB per launch: non-temporal, so that the 3.6 MB of decoder weights the step
        // kernels re-read every step are not evicted from L2 in between
        for (int c = 0; c < NL; ++c) kr[c].u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (size_t)c * KPL * DEC_HD));
#pragma unroll
        for (int c = 0; c < NL; ++c) vr[c].u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (size_t)c * KPL * DEC_HD));
        __builtin_amdgcn_sched_barrier(0);                 // every load of the head is in flight before the first is waited for
        if (first) {
            qs[threadIdx.x] = q_finish<QSPLIT>(qa, q1, scale);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            first = false;
        }
        float qv[EPC];
#pragma unroll
        for (int i = 0; i < EPC; i += 4) {
            const float4 v4 = *reinterpret_cast<const float4*>(qs + h * DEC_HD + dl * EPC + i);
            qv[i] = v4.x; qv[i + 1] = v4.y; qv[i + 2] = v4.z; qv[i + 3] = v4.w;
        }
        float s[NL], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < EPC; ++i) d = fmaf(qv[i], to_f32(kr[c].e[i]), d);
            d = dpp_add<0xB1>(d);                              // quad: lanes of one key row (LPR = 4) ...
            d = dpp_add<0x4E>(d);
            if constexpr (LPR == 8) d = dpp_add<0x141>(d);     // ... or two quads (row_half_mirror)
            s[c] = d;
            mx = fmaxf(mx, d);
        }
        mx = over_keys_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NL; ++c) { s[c] = expf(s[c] - mx); sum += s[c]; }
        const float inv = 1.0f / over_keys_sum(sum);
        float acc[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) acc[i] = 0.f;
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            const float pc = s[c] * inv;
#pragma unroll
            for (int i = 0; i < EPC; ++i) acc[i] = fmaf(pc, to_f32(vr[c].e[i]), acc[i]);
        }
        Piece o;
#pragma unroll
        for (int i = 0; i < EPC; ++i) o.e[i] = from_f32<T>(over_keys_sum(acc[i]));
        if (kl == 0) *reinterpret_cast<u32x4*>(out + (size_t)b * E + h * DEC_HD + dl * EPC) = o.u;
    }
}

// ---- 24-bit K / V rows (bf16x3 mode, PARSeq-S geometry) -----------------------------------------------------------------------------
// The cross-attention is the HBM stream of the memory's K and V (26 times per forward), and the 1e-3 logit tolerance needs 16
// significant bits of them, not 24 (profiles/r04_cheap_exact_study.md: 1.6e-5 on the logits; fp16 rows cost 6e-4, bf16 rows 4e-3).
// Storage: the f32 value rounded to 16 significant bits, bits 31..16 in a u16 plane [B][H][128][32] and bits 15..8 in a u8 plane of
// the same shape `plane_elems` elements behind it — 3 bytes per element, rebuilt with ONE v_perm_b32 each.  Written by
// encoder_blocks_x3.h kv_phase; plans whose K / V come from the generic GEMM (encode() + decode(), parseq_set_memory) keep f32 rows.
__device__ __forceinline__ float f24_even(unsigned hw, unsigned lw, int i) {      // element i (0, 2, 4, 6 of a piece): hi in hw[15:0]
    return __uint_as_float(__builtin_amdgcn_perm(hw, lw, 0x0504000cu | ((unsigned)(i & 3) << 8)));
}
__device__ __forceinline__ float f24_odd(unsigned hw, unsigned lw, int i) {       // element i (1, 3, 5, 7): hi in hw[31:16]
    return __uint_as_float(__builtin_amdgcn_perm(hw, lw, 0x0706000cu | ((unsigned)(i & 3) << 8)));
}
// the 8 elements of a (16-byte u16, 8-byte u8) piece pair
__device__ __forceinline__ void f24_unpack8(const u32x4& hi, const uint2& lo, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const unsigned lw = i < 4 ? lo.x : lo.y;
        v[i] = f24_even(hi[i >> 1], lw, i);
        v[i + 1] = f24_odd(hi[i >> 1], lw, i + 1);
    }
}

// dec_cross_attn_ar_kernel on 24-bit rows: lane (kl = lane / 4, dl = lane % 4) takes elements 8 dl .. 8 dl + 7 of key row 16 c + kl —
// one 16-byte and one 8-byte load per piece, a KiB and half a KiB contiguous per wave load; all 32 loads of a head in flight first.
template <int E, bool QSPLIT = false>
__global__ __launch_bounds__(E)
void dec_cross_attn_ar24_kernel(const float* __restrict__ qc, const QAsm qa, const unsigned char* __restrict__ kmem, const unsigned char* __restrict__ vmem,
                                size_t plane_elems, float scale, float* __restrict__ out) {
    constexpr int H = E / DEC_HD, NK = 128, LPR = 4, KPL = 16, NL = 8;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = E / 64, b = blockIdx.x;
    const int kl = lane / LPR, dl = lane % LPR;
    auto over_keys_sum = [](float v) {
        v = dpp_add<0x124>(v);     // row_ror:4
        v = dpp_add<0x128>(v);     // row_ror:8
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
    auto over_keys_max = [](float v) {
        v = dpp_max<0x124>(v);
        v = dpp_max<0x128>(v);
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        return v;
    };
    __shared__ __attribute__((aligned(16))) float qs[E];
    QOne<QSPLIT> q1;
    q_issue<QSPLIT>(qa, qc, b, E, (int)threadIdx.x, q1);
    bool first = true;
    for (int h = wid; h < H; h += nw) {              // exactly two heads per wave (H = 2 nw): the barrier below is uniform
        const size_t at = (((size_t)b * H + h) * NK + kl) * DEC_HD + dl * 8;
        const unsigned char* kh = kmem + at * 2; const unsigned char* kq = kmem + plane_elems * 2 + at;
        const unsigned char* vh = vmem + at * 2; const unsigned char* vq = vmem + plane_elems * 2 + at;
        u32x4 khi[NL], vhi[NL];
        uint2 klo[NL], vlo[NL];
#pragma unroll
        for (int c = 0; c < NL; ++c) khi[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kh + (size_t)c * KPL * DEC_HD * 2));
#pragma unroll
        for (int c = 0; c < NL; ++c) { const unsigned long long t = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(kq + (size_t)c * KPL * DEC_HD)); klo[c] = make_uint2((unsigned)t, (unsigned)(t >> 32)); }
#pragma unroll
        for (int c = 0; c < NL; ++c) vhi[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vh + (size_t)c * KPL * DEC_HD * 2));
#pragma unroll
        for (int c = 0; c < NL; ++c) { const unsigned long long t = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(vq + (size_t)c * KPL * DEC_HD)); vlo[c] = make_uint2((unsigned)t, (unsigned)(t >> 32)); }
        __builtin_amdgcn_sched_barrier(0);                 // every load of the head is in flight before the first is waited for
        if (first) {
            qs[threadIdx.x] = q_finish<QSPLIT>(qa, q1, scale);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            first = false;
        }
        float qv[8];
#pragma unroll
        for (int i = 0; i < 8; i += 4) {
            const float4 v4 = *reinterpret_cast<const float4*>(qs + h * DEC_HD + dl * 8 + i);
            qv[i] = v4.x; qv[i + 1] = v4.y; qv[i + 2] = v4.z; qv[i + 3] = v4.w;
        }
        float s[NL], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            float kv[8];
            f24_unpack8(khi[c], klo[c], kv);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d = fmaf(qv[i], kv[i], d);
            d = dpp_add<0xB1>(d);                              // quad: the four lanes of one key row
            d = dpp_add<0x4E>(d);
            s[c] = d;
            mx = fmaxf(mx, d);
        }
        mx = over_keys_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NL; ++c) { s[c] = expf(s[c] - mx); sum += s[c]; }
        const float inv = 1.0f / over_keys_sum(sum);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            const float pc = s[c] * inv;
            float vv[8];
            f24_unpack8(vhi[c], vlo[c], vv);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(pc, vv[i], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = over_keys_sum(acc[i]);
        if (kl == 0) {
            float* o = out + (size_t)b * E + h * DEC_HD + dl * 8;
            *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

// Multi-query cross-attention (refinement / NAR): one workgroup of 128 threads per (image, head), Lq <= 32 queries.
// qc fp32 [B*Lq][E]; out T [B*Lq][E].
template <typename T>
__global__ __launch_bounds__(128)
void dec_cross_attn_multi_kernel(const float* __restrict__ qc, const T* __restrict__ kmem, const T* __restrict__ vmem,
                                 int H, int Lq, float scale, T* __restrict__ out) {
    constexpr int NK = 128, QP = DEC_MAXL;
    __shared__ float sk[NK][DEC_HD + 1];        // K rows, +1 pad: thread-per-key row reads hit distinct banks
    __shared__ float sv[DEC_HD][NK + 1];        // V^T rows
    __shared__ float sq[QP][DEC_HD];
    __shared__ float sp[QP][NK + 1];
    const int t = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H, E = H * DEC_HD;
    const T* kg = kmem + (size_t)bh * NK * DEC_HD;
    const T* vg = vmem + (size_t)bh * NK * DEC_HD;
    for (int i = t; i < NK * DEC_HD; i += 128) {
        sk[i / DEC_HD][i % DEC_HD] = to_f32(kg[i]);
        sv[i % DEC_HD][i / DEC_HD] = to_f32(vg[i]);        // V rows arrive [key][d]; the mix wants d-major
    }
    for (int i = t; i < Lq * DEC_HD; i += 128) {
        const int qi = i / DEC_HD, d = i - qi * DEC_HD;
        sq[qi][d] = qc[((size_t)b * Lq + qi) * E + h * DEC_HD + d] * scale;
    }
    __syncthreads();
    {   // scores: thread = key, its K row in registers, queries broadcast from LDS
        float kr[DEC_HD];
#pragma unroll
        for (int d = 0; d < DEC_HD; ++d) kr[d] = sk[t][d];
        for (int qi = 0; qi < Lq; ++qi) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DEC_HD; ++d) s = fmaf(sq[qi][d], kr[d], s);
            sp[qi][t] = s;
        }
    }
    __syncthreads();
    {   // soft-max per query over 128 keys: wave w takes queries w, w + 2
        const int lane = t & 63, wid = t >> 6;
        for (int qi = wid; qi < Lq; qi += 2) {
            const float a = sp[qi][lane], c = sp[qi][lane + 64];
            const float mx = wave_max(fmaxf(a, c));
            const float pa = expf(a - mx), pc = expf(c - mx);
            const float inv = 1.0f / wave_sum(pa + pc);
            sp[qi][lane] = pa * inv; sp[qi][lane + 64] = pc * inv;
        }
    }
    __syncthreads();
    {   // value mix: thread = (d = t & 31, query group g = t >> 5): queries g, g + 4, ...
        const int d = t & 31, g = t >> 5;
        for (int qi = g; qi < Lq; qi += 4) {
            float acc = 0.f;
#pragma unroll 8
            for (int key = 0; key < NK; ++key) acc = fmaf(sp[qi][key], sv[d][key], acc);
            out[((size_t)b * Lq + qi) * E + h * DEC_HD + d] = from_f32<T>(acc);
        }
    }
}

// Cross-attention against any number of memory tokens (row N4: 196 for parseq-patch16-224), 1 <= Lq <= 32 queries per
// (image, head): dec_cross_attn_multi_kernel with a run-time key count and dynamic LDS.  Correctness path for NK != 128.
// LDS floats: NK * 33 (K) + 32 * (NK + 1) (V^T) + 32 * 32 (q) + 32 * (NK + 1) (p).
inline size_t dec_cross_attn_generic_lds(int NK) { return sizeof(float) * ((size_t)NK * (DEC_HD + 1) + 2 * (size_t)DEC_MAXL * (NK + 1) + DEC_MAXL * DEC_HD); }
template <typename T>
__global__ __launch_bounds__(128)
void dec_cross_attn_generic_kernel(const float* __restrict__ qc, const T* __restrict__ kmem, const T* __restrict__ vmem,
                                   int H, int Lq, int NK, float scale, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cag[];
    const int KP = DEC_HD + 1, NP = NK + 1;
    float* sk = reinterpret_cast<float*>(smem_cag);         // [NK][KP]
    float* sv = sk + (size_t)NK * KP;                        // [32][NP]  (d-major)
    float* sp = sv + (size_t)DEC_HD * NP;                    // [32][NP]
    float* sq = sp + (size_t)DEC_MAXL * NP;                  // [32][32]
    const int t = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H, E = H * DEC_HD;
    const T* kg = kmem + (size_t)bh * NK * DEC_HD;
    const T* vg = vmem + (size_t)bh * NK * DEC_HD;
    for (int i = t; i < NK * DEC_HD; i += 128) {
        sk[(i / DEC_HD) * KP + (i % DEC_HD)] = to_f32(kg[i]);
        sv[(i % DEC_HD) * NP + (i / DEC_HD)] = to_f32(vg[i]);
    }
    for (int i = t; i < Lq * DEC_HD; i += 128) {
        const int qi = i / DEC_HD, d = i - qi * DEC_HD;
        sq[qi * DEC_HD + d] = qc[((size_t)b * Lq + qi) * E + h * DEC_HD + d] * scale;
    }
    __syncthreads();
    for (int key = t; key < NK; key += 128) {
        float kr[DEC_HD];
#pragma unroll
        for (int d = 0; d < DEC_HD; ++d) kr[d] = sk[key * KP + d];
        for (int qi = 0; qi < Lq; ++qi) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DEC_HD; ++d) s = fmaf(sq[qi * DEC_HD + d], kr[d], s);
            sp[qi * NP + key] = s;
        }
    }
    __syncthreads();
    {   // soft-max per query: wave w takes queries w, w + 2, ...; lanes stride the keys
        const int lane = t & 63, wid = t >> 6;
        for (int qi = wid; qi < Lq; qi += 2) {
            float mx = -INFINITY;
            for (int key = lane; key < NK; key += 64) mx = fmaxf(mx, sp[qi * NP + key]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int key = lane; key < NK; key += 64) { const float e = expf(sp[qi * NP + key] - mx); sp[qi * NP + key] = e; sum += e; }
            const float inv = 1.0f / wave_sum(sum);
            for (int key = lane; key < NK; key += 64) sp[qi * NP + key] *= inv;
        }
    }
    __syncthreads();
    {
        const int d = t & 31, g = t >> 5;
        for (int qi = g; qi < Lq; qi += 4) {
            float acc = 0.f;
            for (int key = 0; key < NK; ++key) acc = fmaf(sp[qi * NP + key], sv[d * NP + key], acc);
            out[((size_t)b * Lq + qi) * E + h * DEC_HD + d] = from_f32<T>(acc);
        }
    }
}

// Multi-query cross-attention on the matrix cores (bf16 storage): one WAVE per (image, head), Lq <= 32 queries.
//   S^T[key][query] = K Q^T   (16x16x32 MFMA: 8 key tiles x 2 query tiles, the 32-wide head is exactly one k-step)
//   soft-max over the 128 keys of a query: 32 values in the lane + the other three lane groups (two shuffles)
//   O^T[d][query]   = V^T P^T (k-slot s of lane group g in k-step kk is key 32 kk + 16 (s >> 2) + 4 g + (s & 3), which is
//                     where the S^T accumulators already hold the probabilities; V^T fragments are gathered from an LDS copy)
// Queries and probabilities are fp32 quantities in this decoder: each is fed to the bf16 MFMA as a
// hi + lo pair (x = bf16(x) + bf16(x - bf16(x)), two MFMAs), which keeps ~16 mantissa bits instead of 8.
static __global__ __launch_bounds__(256)
void dec_cross_attn_multi_mfma_kernel(const float* __restrict__ qc, const bf16_t* __restrict__ kmem, const bf16_t* __restrict__ vmem,
                                      int H, int Lq, float scale, bf16_t* __restrict__ out, int BH) {
    constexpr int NK = 128, VP = DEC_HD + 2;               // LDS pitch of a V row (elements): odd word count spreads banks
    __shared__ __attribute__((aligned(16))) bf16_t sv[4][NK * VP];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x * 4 + wid;
    if (bh >= BH) return;
    const int b = bh / H, h = bh - b * H, E = H * DEC_HD;
    const bf16_t* kg = kmem + (size_t)bh * NK * DEC_HD;
    const bf16_t* vg = vmem + (size_t)bh * NK * DEC_HD;
    // K fragments straight from global (a key tile is 16 rows x 64 B = one contiguous KB); V staged to LDS
    Frag<bf16_t> kf[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) kf[kt].v = *reinterpret_cast<const bf16x8*>(kg + (kt * 16 + r16) * DEC_HD + 8 * g);
    bf16x8 vraw[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) vraw[c] = *reinterpret_cast<const bf16x8*>(vg + (size_t)c * 512 + lane * 8);
    // queries: lane (query = 16 qt + r16, g) takes d = 8 g .. 8 g + 7, scaled, split hi / lo
    Frag<bf16_t> qhi[2], qlo[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = 16 * qt + r16;
        float qv[8];
        if (qi < Lq) {
            const float4* src = reinterpret_cast<const float4*>(qc + ((size_t)b * Lq + qi) * E + h * DEC_HD + 8 * g);
            const float4 x0 = src[0], x1 = src[1];
            qv[0] = x0.x; qv[1] = x0.y; qv[2] = x0.z; qv[3] = x0.w; qv[4] = x1.x; qv[5] = x1.y; qv[6] = x1.z; qv[7] = x1.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = qv[i] * scale;
            const bf16_t hi = from_f32<bf16_t>(x);
            qhi[qt].v[i] = hi;
            qlo[qt].v[i] = from_f32<bf16_t>(x - to_f32(hi));
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {                       // key row 16 c + lane / 4, piece lane % 4
        bf16_t* dst = &sv[wid][(16 * c + (lane >> 2)) * VP + 8 * (lane & 3)];
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = vraw[c][i];
    }
    f32x4 accs[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            accs[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(accs[qt][kt], kf[kt], qhi[qt]);
            mma16(accs[qt][kt], kf[kt], qlo[qt]);
        }
    // soft-max per query (lane column): 32 keys here, the rest in lanes l ^ 16, l ^ 32, l ^ 48
    Frag<bf16_t> phi[2][4], plo[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, accs[qt][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = expf(accs[qt][kt][r] - mx); accs[qt][kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const float pv = accs[qt][2 * kk + (sl >> 2)][sl & 3] * inv;
                const bf16_t hi = from_f32<bf16_t>(pv);
                phi[qt][kk].v[sl] = hi;
                plo[qt][kk].v[sl] = from_f32<bf16_t>(pv - to_f32(hi));
            }
    }
    __builtin_amdgcn_wave_barrier();                    // this wave's V tile is in LDS (wave-private region)
    f32x4 acco[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) acco[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            Frag<bf16_t> vt;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) vt.v[sl] = sv[wid][(32 * kk + 16 * (sl >> 2) + 4 * g + (sl & 3)) * VP + 16 * dt + r16];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                mma16(acco[qt][dt], vt, phi[qt][kk]);
                mma16(acco[qt][dt], vt, plo[qt][kk]);
            }
        }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = 16 * qt + r16;
        if (qi < Lq) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float o[4] = {acco[qt][dt][0], acco[qt][dt][1], acco[qt][dt][2], acco[qt][dt][3]};
                store4<bf16_t>(out + ((size_t)b * Lq + qi) * E + h * DEC_HD + 16 * dt + 4 * g, o);
            }
        }
    }
}

// bf16x3 form of dec_cross_attn_multi_mfma_kernel: K, V and the output are f32 (the bf16x3 mode's storage); every MFMA operand
// is a bf16 pair (hi, lo = value - hi) and every product three MFMAs (hi lo + lo hi + hi hi, fp32 accumulate) — K and V are split
// as they are loaded, queries and probabilities exactly as in the bf16 kernel.  Two waves per workgroup (the V tile takes two
// LDS planes per wave).  Replaces the VALU kernel dec_cross_attn_multi_kernel<float> for 128 memory tokens (250 -> ~60 us).
// F24: K / V are the 24-bit rows described above (kmem / vmem point at the u16 planes, the u8 planes sit plane_elems elements behind).
template <bool F24>
__global__ __launch_bounds__(128)
void dec_cross_attn_multi_mfma_x3_kernel(const float* __restrict__ qc, const float* __restrict__ kmem, const float* __restrict__ vmem,
                                         int H, int Lq, float scale, float* __restrict__ out, int BH, size_t plane_elems) {
    constexpr int NK = 128, VP = DEC_HD + 2, WAVES = 2;
    __shared__ __attribute__((aligned(16))) bf16_t sv[WAVES][2][NK * VP];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x * WAVES + wid;
    if (bh >= BH) return;
    const int b = bh / H, h = bh - b * H, E = H * DEC_HD;
    const float* kg = kmem + (size_t)bh * NK * DEC_HD;
    const float* vg = vmem + (size_t)bh * NK * DEC_HD;
    // V: 16-byte piece p = 64 c + lane of the [key][32] tile -> key = p >> 3, d = 4 (p & 7); all loads issued first
    // (24-bit rows: 8 elements per piece pair, p = 64 c + lane -> key = p >> 2, d = 8 (p & 3), c < 8)
    f32x4 vraw[16];
    if constexpr (F24) {
        const unsigned char* vh = reinterpret_cast<const unsigned char*>(vmem) + (size_t)bh * NK * DEC_HD * 2;
        const unsigned char* vq = reinterpret_cast<const unsigned char*>(vmem) + plane_elems * 2 + (size_t)bh * NK * DEC_HD;
        u32x4 hi[8]; uint2 lo[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) hi[c] = *reinterpret_cast<const u32x4*>(vh + (size_t)(64 * c + lane) * 16);
#pragma unroll
        for (int c = 0; c < 8; ++c) lo[c] = *reinterpret_cast<const uint2*>(vq + (size_t)(64 * c + lane) * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v[8];
            f24_unpack8(hi[c], lo[c], v);
            vraw[2 * c] = f32x4{v[0], v[1], v[2], v[3]}; vraw[2 * c + 1] = f32x4{v[4], v[5], v[6], v[7]};
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) vraw[c] = *reinterpret_cast<const f32x4*>(vg + (size_t)(64 * c + lane) * 4);
    }
    // K fragments: lane (key = 16 kt + r16, g) takes d = 8 g .. 8 g + 7
    Frag<bf16_t> khi[8], klo[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        f32x4 x0, x1;
        if constexpr (F24) {
            const size_t at = (size_t)bh * NK * DEC_HD + (kt * 16 + r16) * DEC_HD + 8 * g;
            const u32x4 hi = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(kmem) + at * 2);
            const uint2 lo = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(kmem) + plane_elems * 2 + at);
            float v[8];
            f24_unpack8(hi, lo, v);
            x0 = f32x4{v[0], v[1], v[2], v[3]}; x1 = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            const f32x4* src = reinterpret_cast<const f32x4*>(kg + (kt * 16 + r16) * DEC_HD + 8 * g);
            x0 = src[0]; x1 = src[1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t h0 = from_f32<bf16_t>(x0[i]), h1 = from_f32<bf16_t>(x1[i]);
            khi[kt].v[i] = h0; klo[kt].v[i] = from_f32<bf16_t>(x0[i] - to_f32(h0));
            khi[kt].v[4 + i] = h1; klo[kt].v[4 + i] = from_f32<bf16_t>(x1[i] - to_f32(h1));
        }
    }
    Frag<bf16_t> qhi[2], qlo[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = 16 * qt + r16;
        float qv[8];
        if (qi < Lq) {
            const float4* src = reinterpret_cast<const float4*>(qc + ((size_t)b * Lq + qi) * E + h * DEC_HD + 8 * g);
            const float4 x0 = src[0], x1 = src[1];
            qv[0] = x0.x; qv[1] = x0.y; qv[2] = x0.z; qv[3] = x0.w; qv[4] = x1.x; qv[5] = x1.y; qv[6] = x1.z; qv[7] = x1.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = qv[i] * scale;
            const bf16_t hi = from_f32<bf16_t>(x);
            qhi[qt].v[i] = hi;
            qlo[qt].v[i] = from_f32<bf16_t>(x - to_f32(hi));
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        // f32 rows: piece c of the lane; 24-bit rows: half (c & 1) of piece pair c >> 1
        const int key = F24 ? 16 * (c >> 1) + (lane >> 2) : 8 * c + (lane >> 3), d = F24 ? 8 * (lane & 3) + 4 * (c & 1) : 4 * (lane & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t hi = from_f32<bf16_t>(vraw[c][i]);
            sv[wid][0][key * VP + d + i] = hi;
            sv[wid][1][key * VP + d + i] = from_f32<bf16_t>(vraw[c][i] - to_f32(hi));
        }
    }
    f32x4 accs[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            accs[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(accs[qt][kt], khi[kt], qlo[qt]);
            mma16(accs[qt][kt], klo[kt], qhi[qt]);
            mma16(accs[qt][kt], khi[kt], qhi[qt]);
        }
    Frag<bf16_t> phi[2][4], plo[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, accs[qt][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = expf(accs[qt][kt][r] - mx); accs[qt][kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const float pv = accs[qt][2 * kk + (sl >> 2)][sl & 3] * inv;
                const bf16_t hi = from_f32<bf16_t>(pv);
                phi[qt][kk].v[sl] = hi;
                plo[qt][kk].v[sl] = from_f32<bf16_t>(pv - to_f32(hi));
            }
    }
    __builtin_amdgcn_wave_barrier();                    // this wave's V planes are in LDS (wave-private region)
    f32x4 acco[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) acco[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            Frag<bf16_t> vh, vl;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int at = (32 * kk + 16 * (sl >> 2) + 4 * g + (sl & 3)) * VP + 16 * dt + r16;
                vh.v[sl] = sv[wid][0][at];
                vl.v[sl] = sv[wid][1][at];
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                mma16(acco[qt][dt], vh, plo[qt][kk]);
                mma16(acco[qt][dt], vl, phi[qt][kk]);
                mma16(acco[qt][dt], vh, phi[qt][kk]);
            }
        }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = 16 * qt + r16;
        if (qi < Lq) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                *reinterpret_cast<f32x4*>(out + ((size_t)b * Lq + qi) * E + h * DEC_HD + 16 * dt + 4 * g) = acco[qt][dt];
        }
    }
}

// dec_cross_attn_multi_mfma_kernel for any key count NK <= 16 * NT16 (row N4: 196 memory tokens for parseq-patch16-224) and
// 1 <= Lq <= 32 queries: keys padded to NT16 tiles of 16 (scores) / to a multiple of 32 (value mix) with the padded keys
// masked out of the soft-max and zero V rows.  One wave per (image, head); V staged in LDS ([NKP][34] bf16 per wave).
template <int NT16>
__global__ __launch_bounds__(128)
void dec_cross_attn_mfma_n_kernel(const float* __restrict__ qc, const bf16_t* __restrict__ kmem, const bf16_t* __restrict__ vmem,
                                  int H, int Lq, int NK, float scale, bf16_t* __restrict__ out, int BH) {
    constexpr int NKP = ((NT16 + 1) / 2) * 32, KK = NKP / 32, VP = DEC_HD + 2, WAVES = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cam[];
    bf16_t* sv_all = reinterpret_cast<bf16_t*>(smem_cam);              // [WAVES][NKP][VP]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x * WAVES + wid;
    if (bh >= BH) return;
    bf16_t* sv = sv_all + (size_t)wid * NKP * VP;
    const int b = bh / H, h = bh - b * H, E = H * DEC_HD;
    const bf16_t* kg = kmem + (size_t)bh * NK * DEC_HD;
    const bf16_t* vg = vmem + (size_t)bh * NK * DEC_HD;
    Frag<bf16_t> kf[NT16];
#pragma unroll
    for (int kt = 0; kt < NT16; ++kt) {
        const int key = min(kt * 16 + r16, NK - 1);                    // padded keys read a valid row; their scores are masked
        kf[kt].v = *reinterpret_cast<const bf16x8*>(kg + (size_t)key * DEC_HD + 8 * g);
    }
    for (int c = lane; c < NKP * 4; c += 64) {                         // 16-byte piece c: key c >> 2, d 8 (c & 3)
        const int key = c >> 2;
        bf16x8 v = {};
        if (key < NK) v = *reinterpret_cast<const bf16x8*>(vg + (size_t)c * 8);
        bf16_t* dst = sv + key * VP + 8 * (c & 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = v[i];
    }
    Frag<bf16_t> qhi[2], qlo[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = 16 * qt + r16;
        float qv[8];
        if (qi < Lq) {
            const float4* src = reinterpret_cast<const float4*>(qc + ((size_t)b * Lq + qi) * E + h * DEC_HD + 8 * g);
            const float4 x0 = src[0], x1 = src[1];
            qv[0] = x0.x; qv[1] = x0.y; qv[2] = x0.z; qv[3] = x0.w; qv[4] = x1.x; qv[5] = x1.y; qv[6] = x1.z; qv[7] = x1.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = qv[i] * scale;
            const bf16_t hi = from_f32<bf16_t>(x);
            qhi[qt].v[i] = hi;
            qlo[qt].v[i] = from_f32<bf16_t>(x - to_f32(hi));
        }
    }
    const int nqt = Lq > 16 ? 2 : 1;
    f32x4 accs[2][2 * KK];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kt = 0; kt < 2 * KK; ++kt) {
            accs[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < NT16 && qt < nqt) {
                mma16(accs[qt][kt], kf[kt < NT16 ? kt : 0], qhi[qt]);
                mma16(accs[qt][kt], kf[kt < NT16 ? kt : 0], qlo[qt]);
            }
        }
    Frag<bf16_t> phi[2][KK], plo[2][KK];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2 * KK; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kt + 4 * g + r >= NK) accs[qt][kt][r] = -INFINITY;
                mx = fmaxf(mx, accs[qt][kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2 * KK; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = expf(accs[qt][kt][r] - mx); accs[qt][kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const float pv = accs[qt][2 * kk + (sl >> 2)][sl & 3] * inv;
                const bf16_t hi = from_f32<bf16_t>(pv);
                phi[qt][kk].v[sl] = hi;
                plo[qt][kk].v[sl] = from_f32<bf16_t>(pv - to_f32(hi));
            }
    }
    __builtin_amdgcn_wave_barrier();
    f32x4 acco[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) acco[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            Frag<bf16_t> vt;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) vt.v[sl] = sv[(32 * kk + 16 * (sl >> 2) + 4 * g + (sl & 3)) * VP + 16 * dt + r16];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                if (qt < nqt) {
                    mma16(acco[qt][dt], vt, phi[qt][kk]);
                    mma16(acco[qt][dt], vt, plo[qt][kk]);
                }
            }
        }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = 16 * qt + r16;
        if (qi < Lq) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float o[4] = {acco[qt][dt][0], acco[qt][dt][1], acco[qt][dt][2], acco[qt][dt][3]};
                store4<bf16_t>(out + ((size_t)b * Lq + qi) * E + h * DEC_HD + 16 * dt + 4 * g, o);
            }
        }
    }
}
template <int NT16> constexpr size_t dec_cross_attn_mfma_n_lds() { return (size_t)2 * (((NT16 + 1) / 2) * 32) * (DEC_HD + 2) * sizeof(bf16_t); }

}  // namespace pq

