// LayerNorm + Linear for the two widest encoder projections (qkv: E -> 3E, fc1: E -> 4E), bf16 throughput mode.
//
//   out[m][n] = epi( LayerNorm(x[m]) . W[n] + bias[n] )          x: fp32 residual stream [M, E];  W: bf16 [N, E]
//
// Why a dedicated kernel (profiles/r01_*: the generic tile kernel spent half of its time per 128x128 tile in fixed costs —
// first-stage load latency, LDS staging of the output — and every n-tile re-read the A rows; the LayerNorm in front was
// a separate 150 MB pass):
//   * A-stationary, in REGISTERS: a workgroup owns 128 rows; each of its 4 waves owns 32 of them and keeps their
//     LayerNorm'd bf16 values as MFMA operand fragments for the whole K = E depth (2 x E/32 x 4 VGPRs = 96 at E = 384).
//     LayerNorm is computed on the way in (shifted two-moment statistics, then a second pass that normalises), so there
//     is no normalised-activation tensor in HBM at all and A is read from L2/HBM once per row, not once per n-tile.
//   * W-streaming: the workgroup walks ALL N/128 column tiles; W tiles stream through a 4-slot LDS ring of 16 KiB
//     stages (128 W rows x 64 k) filled by global_load_lds_dwordx4, three stages in flight across raw s_barriers with
//     counted s_waitcnt vmcnt — the stream never drains between column tiles, so first-load latency is paid once per
//     workgroup instead of once per tile.
//   * No LDS staging of the output: the W rows of a stage are placed in LDS in a permuted order such that, after the
//     MFMAs, each lane holds 16 CONSECUTIVE output columns of one row -> two 16-byte stores per (row tile, column quad),
//     128 contiguous bytes per row across the four lanes that share it.
//
// MFMA operand roles (common.h mma16): first operand = W fragment (rows n), second = A fragment (rows m); lane l then
// holds D[n = 4 (l >> 4) + r][m = l & 15].
//
// Wait-count discipline: a wave issues 4 DMA instructions per stage and keeps at most 3 stages (12) in flight.  Before
// reading stage s it waits until at most 8 VMEM operations are outstanding.  Stores of the previous column tile may be
// outstanding too and may retire out of order with respect to loads; since loads retire in order among themselves,
// "at most 8 outstanding" still implies the 4 oldest loads (stage s) have landed — stores can only make the wait
// conservative, never unsafe.
#pragma once
#include "common.h"

namespace pq {

constexpr int PN_BM = 128, PN_BN = 128, PN_NST = 4, PN_STAGE_BYTES = PN_BN * 128;

// Epilogues: store16(m, n, v) receives 16 consecutive output columns n .. n+15 of row m (bias already added).
struct PanelHeads {          // q, k, v all as [b][h][t][d] (row-major per head), hd = 64
    bf16_t* seg[3]; int E, heads, hd, tokens;
    __device__ __forceinline__ void store16(int m, int n, const float* v) const {
        const int which = n / E, col = n - which * E;
        const int h = col / hd, d = col - h * hd;
        const int b_ = m / tokens, t = m - b_ * tokens;
        bf16_t* p = seg[which] + (((size_t)b_ * heads + h) * tokens + t) * hd + d;
        bf16x8 lo, hi;
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo[i] = static_cast<bf16_t>(v[i]); hi[i] = static_cast<bf16_t>(v[8 + i]); }
        *reinterpret_cast<bf16x8*>(p) = lo;
        *reinterpret_cast<bf16x8*>(p + 8) = hi;
    }
};

struct PanelGelu {           // out[m][n] = gelu(.), row-major [M, N]
    bf16_t* out; int ldo;
    __device__ __forceinline__ void store16(int m, int n, const float* v) const {
        bf16_t* p = out + (size_t)m * ldo + n;
        bf16x8 lo, hi;
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo[i] = static_cast<bf16_t>(gelu_erf(v[i])); hi[i] = static_cast<bf16_t>(gelu_erf(v[8 + i])); }
        *reinterpret_cast<bf16x8*>(p) = lo;
        *reinterpret_cast<bf16x8*>(p + 8) = hi;
    }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int E, typename Epi>
__global__ __launch_bounds__(256, 2)
void ln_panel_gemm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                          float eps, const bf16_t* __restrict__ W, const float* __restrict__ bias, int M, int N,
                          const Epi epi) {
    constexpr int KSTEPS = E / 32;            // MFMA k-steps over the full depth
    constexpr int KS = E / 64;                // 128-byte stages per column tile
    static_assert(E % 64 == 0, "E must be a multiple of 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                                             // [PN_NST][128 rows][128 B], XOR-swizzled
    float* sbias = reinterpret_cast<float*>(smem + PN_NST * PN_STAGE_BYTES);  // [N]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * PN_BM;
    const int ntiles = N / PN_BN, S = ntiles * KS;

    // ---- W stream: stage s = (column tile s / KS, k-stage s % KS) -> ring slot s % 4 --------------------------------
    // LDS row rho = i * 16 + r16 of a stage holds W row n0 + p(rho),  p = (i >> 2) * 64 + (r16 >> 2) * 16 + (i & 3) * 4 + (r16 & 3)
    // so that MFMA tile i, accumulator register r of lane group g is output column n0 + (i >> 2) * 64 + 16 g + (i & 3) * 4 + r.
    // One DMA instruction = 8 LDS rows; wave w issues the 4 instructions covering LDS rows 32 w .. 32 w + 31.
    int w_rowoff[4];                          // element offset of this lane's source row inside a column tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rho = (wid * 4 + q) * 8 + (lane >> 3);
        const int i = rho >> 4, r16 = rho & 15;
        const int p = (i >> 2) * 64 + (r16 >> 2) * 16 + (i & 3) * 4 + (r16 & 3);
        w_rowoff[q] = p * E + (((lane & 7) ^ (rho & 7)) * 8);
    }
    auto issue_stage = [&](int s) {
        const int nt = s / KS, kt = s - nt * KS;
        const bf16_t* base = W + (size_t)nt * PN_BN * E + kt * 64;
        unsigned char* slot = ring + (s & (PN_NST - 1)) * PN_STAGE_BYTES + wid * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + w_rowoff[q]),
                                             (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
    };
    issue_stage(0);
    if (S > 1) issue_stage(1);
    if (S > 2) issue_stage(2);

    for (int i = tid; i < N; i += 256) sbias[i] = bias[i];

    // ---- A panel: LayerNorm'd rows of this wave as MFMA fragments in registers --------------------------------------
    // lane (r16, g) of row tile j holds row m0 + 32 wid + 16 j + r16, elements [32 ks + 8 g, +8) for every k-step ks.
    bf16x8 afrag[2][KSTEPS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = min(m0 + wid * 32 + j * 16 + rr, M - 1);
        const float* xr = x + (size_t)row * E + 8 * g;
        const float c = x[(size_t)row * E];                  // shift for the one-pass moments (any sample of the row)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(xr + ks * 32);
            const float4 b = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
            const float d0 = a.x - c, d1 = a.y - c, d2 = a.z - c, d3 = a.w - c, d4 = b.x - c, d5 = b.y - c, d6 = b.z - c, d7 = b.w - c;
            s1 += ((d0 + d1) + (d2 + d3)) + ((d4 + d5) + (d6 + d7));
            s2 += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
        }
        // the row is spread over the 4 lane groups: lanes r16, r16 + 16, r16 + 32, r16 + 48
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        const float dm = s1 * (1.0f / E);                    // mean - c
        const float var = fmaxf(s2 * (1.0f / E) - dm * dm, 0.f);
        const float mean = c + dm, rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(xr + ks * 32);
            const float4 b = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
            const float4 ga = *reinterpret_cast<const float4*>(gamma + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(gamma + ks * 32 + 8 * g + 4);
            const float4 ba = *reinterpret_cast<const float4*>(beta + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(beta + ks * 32 + 8 * g + 4);
            bf16x8 f;
            f[0] = static_cast<bf16_t>((a.x - mean) * rstd * ga.x + ba.x); f[1] = static_cast<bf16_t>((a.y - mean) * rstd * ga.y + ba.y);
            f[2] = static_cast<bf16_t>((a.z - mean) * rstd * ga.z + ba.z); f[3] = static_cast<bf16_t>((a.w - mean) * rstd * ga.w + ba.w);
            f[4] = static_cast<bf16_t>((b.x - mean) * rstd * gb.x + bb.x); f[5] = static_cast<bf16_t>((b.y - mean) * rstd * gb.y + bb.y);
            f[6] = static_cast<bf16_t>((b.z - mean) * rstd * gb.z + bb.z); f[7] = static_cast<bf16_t>((b.w - mean) * rstd * gb.w + bb.w);
            afrag[j][ks] = f;
        }
    }

    // ---- main loop over the W stream ---------------------------------------------------------------------------------
    f32x4 acc[8][2];
    const int sx = rr & 7;
    const int frag_off = rr * 128;
    for (int nt = 0; nt < ntiles; ++nt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kt = 0; kt < KS; ++kt) {
            const int s = nt * KS + kt;
            // stage s has landed for this wave once <= 4 * min(2, S-1-s) VMEM ops are outstanding (see header)
            if (s + 2 < S) wait_vmcnt<8>(); else if (s + 1 < S) wait_vmcnt<4>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // everyone's share of stage s is in LDS; slot (s+3)%4 is free again
            asm volatile("" ::: "memory");
            if (s + 3 < S) issue_stage(s + 3);
            const unsigned char* st = ring + (s & (PN_NST - 1)) * PN_STAGE_BYTES + frag_off;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int so = ((kk * 4 + g) ^ sx) * 16;
                bf16x8 wf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + i * 2048 + so);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], afrag[0][kt * 2 + kk], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], afrag[1][kt * 2 + kk], acc[i][1], 0, 0, 0);
                }
            }
        }
        // ---- column-tile epilogue straight from registers: 16 consecutive columns per lane ------------------------
        const int n0 = nt * PN_BN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wid * 32 + j * 16 + rr;
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                const int n = n0 + qd * 64 + 16 * g;
                float v[16];
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[i4 * 4 + r] = acc[qd * 4 + i4][j][r] + sbias[n + i4 * 4 + r];
                if (m < M) epi.store16(m, n, v);
            }
        }
    }
}

template <int E, typename Epi>
inline hipError_t launch_ln_panel_gemm(hipStream_t s, const float* x, const float* gamma, const float* beta, float eps,
                                       const bf16_t* W, const float* bias, int M, int N, const Epi& epi) {
    const size_t lds = (size_t)PN_NST * PN_STAGE_BYTES + (size_t)N * sizeof(float);
    auto kern = ln_panel_gemm_kernel<E, Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((M + PN_BM - 1) / PN_BM), dim3(256), lds, s, x, gamma, beta, eps, W, bias, M, N, epi);
    return hipGetLastError();
}

}  // namespace pq
