// Decoder attention kernels (nn.MultiheadAttention semantics), restructured for the
// depth-1 two-stream decoder:
//
//  * The content stream of a 1-layer decoder is a pure function of (position, token id), so its LayerNorm'd K/V
//    projection is a lookup table kvtab[pos][tok][2E] built once per weight set (K in [0,E), V in [E,2E)).
//    The position queries are batch-independent too: qself[pos][E] = (Wq norm_q(pos_queries[pos]) + bq) / sqrt(hd).
//    dec_self_attn_kernel therefore only gathers, dots, soft-maxes and mixes — no GEMM in the AR step's self-attention.
//  * Cross-attention K/V of `memory` are projected ONCE per image (kvmem[b*Nk + key][2E]); the reference re-projects
//    them on each of its 26 + refine_iters calls.
//
// Layout trick shared by both kernels: lane l of a wave owns feature (64 s + l) of "stripe" s; with hd = 32 a stripe
// holds exactly two heads (lanes 0-31 and 32-63), so a head's q.k dot product is a 32-lane butterfly sum and the
// soft-max / value mixing are lane-local.
#pragma once
#include "common.h"

namespace pq {

constexpr int DEC_MAXL = 26 + 6;   // max context length supported by the register-resident score array

// One wave per (image b, query index qi).  Query position = i0 + qi.  Keys j = 0 .. Lk-1 are the content tokens
// tok[b][j].  Masks follow torch: qmask[(i0 + qi) * ldq + j] != 0  or  kpm[b * ldk + j] != 0  => key j is -inf.
template <typename T, int E>
__global__ __launch_bounds__(256)
void dec_self_attn_kernel(const float* __restrict__ qself, const T* __restrict__ kvtab, const int* __restrict__ tok,
                          int ldt, int ntok, const unsigned char* __restrict__ qmask, int ldq,
                          const unsigned char* __restrict__ kpm, int ldk, int Lk, int i0, int Lq, int B,
                          T* __restrict__ out) {
    constexpr int NS = E / 64;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= B * Lq) return;
    const int b = w / Lq, qi = w - b * Lq, pos = i0 + qi;

    int mytok = 0; bool mymasked = true;
    if (lane < Lk) {
        mytok = tok[(size_t)b * ldt + lane];
        mymasked = (qmask && qmask[(size_t)pos * ldq + lane]) || (kpm && kpm[(size_t)b * ldk + lane]);
    }
    const unsigned long long masked = __ballot(mymasked);

#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
        const int f = s * 64 + lane;
        const float qv = qself[(size_t)pos * E + f];
        float sc[DEC_MAXL];
        float vv[DEC_MAXL];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < DEC_MAXL; ++j) {
            sc[j] = -INFINITY; vv[j] = 0.f;
            if (j < Lk) {
                const int tj = __shfl(mytok, j, 64);
                const T* rowp = kvtab + ((size_t)j * ntok + tj) * (2 * E) + f;
                const float kval = to_f32(rowp[0]);
                vv[j] = to_f32(rowp[E]);
                const float d = half_sum(qv * kval);
                if (!((masked >> j) & 1ull)) { sc[j] = d; mx = fmaxf(mx, d); }
            }
        }
        float sum = 0.f, acc = 0.f;
#pragma unroll
        for (int j = 0; j < DEC_MAXL; ++j) {
            if (j < Lk) {
                const float p = expf(sc[j] - mx);     // exp(-inf) = 0 for masked keys
                sum += p;
                acc = fmaf(p, vv[j], acc);
            }
        }
        out[(size_t)w * E + f] = from_f32<T>(acc / sum);
    }
}

// Cross-attention of QC queries of one image against its Nk memory tokens.  Workgroup = 4 waves; wave w scans keys
// [w Nk/4, (w+1) Nk/4) with an online soft-max per (query, head) and the four partial states are merged through LDS
// (flash-decoding style) — the AR step has only one query per image, so the key split is where its parallelism comes
// from.  qc: fp32 [B*Lq][E] un-scaled projected queries; kvmem: T [B*Nk][2E]; out: T [B*Lq][E].
template <typename T, int E, int QC>
__global__ __launch_bounds__(256)
void dec_cross_attn_kernel(const float* __restrict__ qc, const T* __restrict__ kvmem, int Nk, int Lq, float scale,
                           T* __restrict__ out) {
    constexpr int NS = E / 64;
    __shared__ float part[3][4][QC][64];          // {m, l, acc} x wave x query x lane
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int chunks = (Lq + QC - 1) / QC;
    const int b = blockIdx.x / chunks, c0 = (blockIdx.x - b * chunks) * QC;
    const int kper = Nk / 4, kbeg = wid * kper;

#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
        const int f = s * 64 + lane;
        float qv[QC], m[QC], l[QC], a[QC];
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) {
            const int qi = c0 + qq;
            qv[qq] = (qi < Lq) ? qc[((size_t)b * Lq + qi) * E + f] * scale : 0.f;
            m[qq] = -INFINITY; l[qq] = 0.f; a[qq] = 0.f;
        }
        const T* kp = kvmem + ((size_t)b * Nk + kbeg) * (2 * E) + f;
#pragma unroll 4
        for (int key = 0; key < kper; ++key) {
            const float kval = to_f32(kp[(size_t)key * 2 * E]);
            const float vval = to_f32(kp[(size_t)key * 2 * E + E]);
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
                const float sc = half_sum(qv[qq] * kval);
                const float mn = fmaxf(m[qq], sc);
                const float corr = expf(m[qq] - mn);       // first key: exp(-inf) = 0
                const float p = expf(sc - mn);
                l[qq] = l[qq] * corr + p;
                a[qq] = a[qq] * corr + p * vval;
                m[qq] = mn;
            }
        }
        __syncthreads();                                    // previous stripe's merge has finished reading `part`
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) { part[0][wid][qq][lane] = m[qq]; part[1][wid][qq][lane] = l[qq]; part[2][wid][qq][lane] = a[qq]; }
        __syncthreads();
        for (int qq = wid; qq < QC; qq += 4) {
            const int qi = c0 + qq;
            if (qi >= Lq) continue;
            float M = -INFINITY;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) M = fmaxf(M, part[0][w2][qq][lane]);
            float L = 0.f, A = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const float e = expf(part[0][w2][qq][lane] - M);
                L += part[1][w2][qq][lane] * e;
                A += part[2][w2][qq][lane] * e;
            }
            out[((size_t)b * Lq + qi) * E + f] = from_f32<T>(A / L);
        }
    }
}

}  // namespace pq
