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
// on each of its 26 + refine_iters decoder calls) and stored head-split:  kmem[b][h][key][32], vtmem[b][h][32][key].
//   * AR step (one query per image): dec_cross_attn_ar_kernel — memory-bound streaming of the image's 2 x 96 KB of K/V
//     with 16-byte loads, one workgroup of E threads per image.
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
                          const unsigned char* __restrict__ kpm, int ldk, int Lk, int i0, int Lq, T* __restrict__ out) {
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
        sp[h][j] = masked ? -INFINITY : stab[(((size_t)pos * npos + j) * ntok + stok[j]) * H + h];
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

// AR-step cross-attention: one workgroup of E threads per image, ONE query.  qc fp32 [B][E] un-scaled projected query;
// kmem T [B][H][Nk][32]; vtmem T [B][H][32][Nk]; out T [B][E].  Nk = 128 memory tokens.
template <typename T, int E>
__global__ __launch_bounds__(E)
void dec_cross_attn_ar_kernel(const float* __restrict__ qc, const T* __restrict__ kmem, const T* __restrict__ vtmem,
                              float scale, T* __restrict__ out) {
    constexpr int H = E / DEC_HD, NK = 128;
    constexpr int EPC = 16 / (int)sizeof(T);           // elements per 16-byte chunk
    __shared__ __attribute__((aligned(16))) float sq[E];
    __shared__ __attribute__((aligned(16))) float sp[H][NK];
    const int t = threadIdx.x, b = blockIdx.x;
    sq[t] = qc[(size_t)b * E + t] * scale;
    __syncthreads();
    // scores: H * NK (head, key) pairs, consecutive threads take consecutive keys of one head -> consecutive 64-byte rows
    for (int pair = t; pair < H * NK; pair += E) {
        const int h = pair / NK, key = pair - h * NK;
        const T* kr = kmem + (((size_t)b * H + h) * NK + key) * DEC_HD;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DEC_HD / EPC; ++c) {
            union { u32x4 u; T e[EPC]; } kv;
            kv.u = *reinterpret_cast<const u32x4*>(kr + c * EPC);
#pragma unroll
            for (int i = 0; i < EPC; ++i) s = fmaf(sq[h * DEC_HD + c * EPC + i], to_f32(kv.e[i]), s);
        }
        sp[h][key] = s;
    }
    __syncthreads();
    // soft-max over the 128 keys of each head: wave w handles heads w, w + nwaves, ...; lane owns keys lane and lane + 64
    {
        const int lane = t & 63, wid = t >> 6, nw = E / 64;
        for (int h = wid; h < H; h += nw) {
            const float a = sp[h][lane], c = sp[h][lane + 64];
            const float mx = wave_max(fmaxf(a, c));
            const float pa = expf(a - mx), pc = expf(c - mx);
            const float inv = 1.0f / wave_sum(pa + pc);
            sp[h][lane] = pa * inv; sp[h][lane + 64] = pc * inv;
        }
    }
    __syncthreads();
    // value mix: thread t = (h, d) walks its V^T row (128 keys, contiguous) against the head's probabilities
    {
        const int h = t / DEC_HD;
        const T* vr = vtmem + ((size_t)b * E + t) * NK;          // ((b * H + h) * 32 + d) * NK with h * 32 + d == t
        float acc = 0.f;
#pragma unroll 4
        for (int c = 0; c < NK / EPC; ++c) {
            union { u32x4 u; T e[EPC]; } vv;
            vv.u = *reinterpret_cast<const u32x4*>(vr + c * EPC);
#pragma unroll
            for (int i = 0; i < EPC; ++i) acc = fmaf(sp[h][c * EPC + i], to_f32(vv.e[i]), acc);
        }
        out[(size_t)b * E + t] = from_f32<T>(acc);
    }
}

// Multi-query cross-attention (refinement / NAR): one workgroup of 128 threads per (image, head), Lq <= 32 queries.
// qc fp32 [B*Lq][E]; out T [B*Lq][E].
template <typename T>
__global__ __launch_bounds__(128)
void dec_cross_attn_multi_kernel(const float* __restrict__ qc, const T* __restrict__ kmem, const T* __restrict__ vtmem,
                                 int H, int Lq, float scale, T* __restrict__ out) {
    constexpr int NK = 128, QP = DEC_MAXL;
    __shared__ float sk[NK][DEC_HD + 1];        // K rows, +1 pad: thread-per-key row reads hit distinct banks
    __shared__ float sv[DEC_HD][NK + 1];        // V^T rows
    __shared__ float sq[QP][DEC_HD];
    __shared__ float sp[QP][NK + 1];
    const int t = threadIdx.x;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H, E = H * DEC_HD;
    const T* kg = kmem + (size_t)bh * NK * DEC_HD;
    const T* vg = vtmem + (size_t)bh * DEC_HD * NK;
    for (int i = t; i < NK * DEC_HD; i += 128) {
        sk[i / DEC_HD][i % DEC_HD] = to_f32(kg[i]);
        sv[i / NK][i % NK] = to_f32(vg[i]);
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

}  // namespace pq
