// LayerNorm + Linear for the two widest encoder projections (qkv: E -> 3E, fc1: E -> 4E), bf16 throughput mode.
//
//   out[m][n] = epi( LayerNorm(x[m]) . W[n] + bias[n] )          x: fp32 residual stream [M, E];  W: bf16 [N, E]
//
// Why a dedicated kernel (profiles/r01_*: the generic tile kernel spent half of its time per 128x128 tile in fixed costs —
// first-stage load latency, LDS staging of the output — and every n-tile re-read the A rows; the LayerNorm in front was
// a separate 150 MB pass):
//   * A-stationary, in REGISTERS: a workgroup owns 128 rows; each of its 4 waves owns 32 of them and keeps their
//     LayerNorm'd bf16 values as MFMA operand fragments for the whole K = E depth (2 x E/32 x 4 VGPRs = 96 at E = 384).
//     LayerNorm is computed on the way in (two-pass statistics on the register-resident row slices: x is read exactly once), so there
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
// Wait-count discipline.  A wave issues 4 DMA instructions per stage and keeps 3 stages (12 loads) in flight; the 8
// global stores of a column tile's epilogue are never waited for — they drain under the next tile's MFMAs.  On the
// GFX9 family global (non-FLAT) vector memory operations report completion to a wavefront in execution order, loads
// and stores alike (LLVM AMDGPUUsage, memory model GFX6-GFX9 / GFX942), so "stage s has landed" is exactly
// "at most A VMEM operations issued after stage s's loads are still outstanding", with
//     A = 4 * min(2, stages left after s)  +  8 if the previous tile's stores were issued after stage s's loads
// (true for the first three stages of every column tile but the first).  The first version used A without the store
// term; that is also correct but makes every wave sit out its own store latency once per column tile, and because all
// 512 workgroups hit their epilogues together the stores arrived in 17 MB bursts: 88 us of a 194 us kernel
// (ablations recorded in profiles/r01_panel_ablation.log; the harness that produced them was removed with the ablation variants, commit c0ffb4f).
#pragma once
#include <type_traits>

#include "common.h"

namespace pq {

constexpr int PN_BM = 128, PN_BN = 128, PN_NST = 4, PN_STAGE_BYTES = PN_BN * 128;

// Epilogues.  pack8(v) turns 8 consecutive output columns of one row (bias already added) into a 16-byte bf16 piece;
// store_piece(m, n, piece) writes that piece at row m, columns n .. n+7.  The kernel arranges for one store instruction to
// cover 8 rows x 128 contiguous bytes (see the exchange in the epilogue).
__device__ __forceinline__ u32x4 pack_bf16x8(const float* v) {
    union { u32x4 u; bf16_t e[8]; } o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.e[i] = static_cast<bf16_t>(v[i]);
    return o.u;
}

struct PanelHeads {          // q, k, v all as [b][h][t][d] (row-major per head), hd = 64
    bf16_t* seg[3]; int E, heads, hd, tokens;
    __device__ __forceinline__ u32x4 pack8(const float* v) const { return pack_bf16x8(v); }
    // EC: the kernel's compile-time embedding width (== E); hd is 64.  The divisions by run-time values this function used to
    // carry (n / E, col / hd, m / tokens: sixteen stores per column tile) were a VALU load comparable to the tile's MFMAs.
    template <int EC>
    __device__ __forceinline__ void store_piece(int m, int n, const u32x4& piece) const {
        const int which = n / EC, col = n - which * EC;
        const int h = col >> 6, d = col & 63;
        const int b_ = tokens == 128 ? (m >> 7) : m / tokens, t = m - b_ * tokens;
        *reinterpret_cast<u32x4*>(seg[which] + (((size_t)b_ * heads + h) * tokens + t) * 64 + d) = piece;
    }
};

struct PanelGelu {           // out[m][n] = gelu(.), row-major [M, N]
    bf16_t* out; int ldo;
    __device__ __forceinline__ u32x4 pack8(const float* v) const {
        float g8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) g8[i] = gelu_poly(v[i]);
        return pack_bf16x8(g8);
    }
    template <int EC>
    __device__ __forceinline__ void store_piece(int m, int n, const u32x4& piece) const {
        *reinterpret_cast<u32x4*>(out + (size_t)m * ldo + n) = piece;
    }
};

// swap 16-byte pieces between lane r16 and lane r16 ^ 8 of every row of 16 lanes (DPP row_ror:8)
__device__ __forceinline__ u32x4 swap_half_rows(const u32x4& v) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[i], 0x128, 0xF, 0xF, false);
    return o;
}

#ifndef PN_EARLY
#define PN_EARLY 0        // k-steps of row tile 1 requested together with row tile 0 (measured: 0 -> 94.6 us, 6 -> 98.9, 12 -> 100.4:
                          // the extra live registers turn into prologue spills that cost more than the latency they hide)
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (Ablation variants — no stores, no LayerNorm prologue, no MFMAs, no stream waits: profiles/r01_panel_ablation.log — are in the
// history at commit c342b0c, the parent of the pruning commit c0ffb4f.)
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
    // LDS row rho = i * 16 + r16 of a stage holds W row n0 + p(rho),
    //     p = (i >> 2) * 64 + ((i >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3)
    // so that MFMA tile i, accumulator register r of lane group g is output column
    //     n0 + (i >> 2) * 64 + ((i >> 1) & 1) * 32 + 8 g + (i & 1) * 4 + r:
    // a tile PAIR gives a lane 8 consecutive columns, and the four lane groups that share a row cover 32 consecutive
    // columns = 64 contiguous bytes per row per store instruction (the first layout, 16 columns per lane in two 16-byte
    // halves, left 16-byte holes between lanes inside one instruction and ran the stores at 2.3 TB/s).
    // One DMA instruction = 8 LDS rows; wave w issues the 4 instructions covering LDS rows 32 w .. 32 w + 31.
    int w_rowoff[4];                          // element offset of this lane's source row inside a column tile
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rho = (wid * 4 + q) * 8 + (lane >> 3);
        const int i = rho >> 4, r16 = rho & 15;
        const int p = (i >> 2) * 64 + ((i >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3);
        w_rowoff[q] = p * E + (((lane & 7) ^ (rho & 7)) * 8);
    }
    // Column tiles are walked in a per-workgroup rotated order (tile (i + rot) % ntiles at step i): with every workgroup on
    // the same tile at the same time, the 512 x 128 rows x 256 B written per step sit at a fixed offset inside a 2.3-3 KB row
    // pitch and camp on a fraction of the HBM channels; rotating spreads each step's stores over the whole row width.
    const int rot = blockIdx.x % ntiles;
    auto issue_stage = [&](int s) {
        const int ni = s / KS, kt = s - ni * KS;
        int nt = ni + rot; if (nt >= ntiles) nt -= ntiles;
        const bf16_t* base = W + (size_t)nt * PN_BN * E + kt * 64;
        unsigned char* slot = ring + (s & (PN_NST - 1)) * PN_STAGE_BYTES + wid * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + w_rowoff[q]),
                                             (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
    };
    for (int i = tid; i < N; i += 256) sbias[i] = bias[i];     // before the DMA prefetch (ordinary loads behind it drain it)
    // the first three weight stages are requested right after row tile 0's activation loads (below): vector memory returns in
    // order, so activation rows queued behind twelve LDS-DMA copies would wait for all of them before the LayerNorm can start

    // ---- A panel: LayerNorm'd rows of this wave as MFMA fragments in registers --------------------------------------
    // lane (r16, g) of row tile j holds row m0 + 32 wid + 16 j + r16, elements [32 ks + 8 g, +8) for every k-step ks.
    bf16x8 afrag[2][KSTEPS];
    const bool lo_half = rr < 8;
    // ONE pass over x, fully coalesced: per k-step (128 bytes of a row) the 8 lanes {(r16 & 7, g), ((r16 & 7) + 8, g)}
    // read the 8 consecutive 16-byte pieces of ONE row — lanes r16 < 8 the even pieces 2g, lanes >= 8 the odd pieces
    // 2g+1 — first for rows 0-7, then for rows 8-15; a DPP half-row swap then gives every lane both pieces of its own
    // row.  (Reading each lane's own 32 bytes directly leaves 16-byte holes between lanes inside an instruction.)
    // Row tile 1's first PN_EARLY k-steps can be requested together with row tile 0 (latency behind tile 0's LayerNorm
    // arithmetic) — but at the 256-register budget that costs more in spills than it hides (see PN_EARLY).
    u32x4 raw0[2][KSTEPS], raw1[2][KSTEPS];
    auto load_rows = [&](auto jc, auto k0c, auto k1c) {
        constexpr int j = decltype(jc)::value, k0 = decltype(k0c)::value, k1 = decltype(k1c)::value;
        const int rbase = m0 + wid * 32 + j * 16 + (rr & 7);
        const float* xlo = x + (size_t)min(rbase, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
        const float* xhi = x + (size_t)min(rbase + 8, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
#pragma unroll
        for (int ks = k0; ks < k1; ++ks) {
            raw0[j][ks] = *reinterpret_cast<const u32x4*>(xlo + ks * 32);            // a piece of row (r16 & 7)
            raw1[j][ks] = *reinterpret_cast<const u32x4*>(xhi + ks * 32);            // a piece of row (r16 & 7) + 8
        }
    };
    auto ln_rows = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        float4 xa[KSTEPS], xb[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const u32x4 p0 = raw0[j][ks], p1 = raw1[j][ks];
            // lanes < 8 own the first row: keep p0 (even piece), give p1; lanes >= 8 own the second: keep p1 (odd piece), give p0
            const u32x4 got = swap_half_rows(lo_half ? p1 : p0);
            const u32x4 ev = lo_half ? p0 : got, od = lo_half ? got : p1;          // even piece 2g, odd piece 2g+1 of MY row
            xa[ks] = make_float4(__uint_as_float(ev[0]), __uint_as_float(ev[1]), __uint_as_float(ev[2]), __uint_as_float(ev[3]));
            xb[ks] = make_float4(__uint_as_float(od[0]), __uint_as_float(od[1]), __uint_as_float(od[2]), __uint_as_float(od[3]));
        }
        float s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) s1 += ((xa[ks].x + xa[ks].y) + (xa[ks].z + xa[ks].w)) + ((xb[ks].x + xb[ks].y) + (xb[ks].z + xb[ks].w));
        // the row is spread over the 4 lane groups: lanes r16, r16 + 16, r16 + 32, r16 + 48
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
        const float rstd = __builtin_amdgcn_rsqf(s2 * (1.0f / E) + eps);       // v_rsq_f32, 1 ulp
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const float4 a = xa[ks], b = xb[ks];
            const float4 ga = *reinterpret_cast<const float4*>(gamma + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(gamma + ks * 32 + 8 * g + 4);
            const float4 ba = *reinterpret_cast<const float4*>(beta + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(beta + ks * 32 + 8 * g + 4);
            bf16x8 f;
            f[0] = static_cast<bf16_t>((a.x - mean) * rstd * ga.x + ba.x); f[1] = static_cast<bf16_t>((a.y - mean) * rstd * ga.y + ba.y);
            f[2] = static_cast<bf16_t>((a.z - mean) * rstd * ga.z + ba.z); f[3] = static_cast<bf16_t>((a.w - mean) * rstd * ga.w + ba.w);
            f[4] = static_cast<bf16_t>((b.x - mean) * rstd * gb.x + bb.x); f[5] = static_cast<bf16_t>((b.y - mean) * rstd * gb.y + bb.y);
            f[6] = static_cast<bf16_t>((b.z - mean) * rstd * gb.z + bb.z); f[7] = static_cast<bf16_t>((b.w - mean) * rstd * gb.w + bb.w);
            afrag[j][ks] = f;
        }
    };
    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using KE = std::integral_constant<int, PN_EARLY>; using KA = std::integral_constant<int, KSTEPS>;
        load_rows(I0{}, I0{}, KA{});
        load_rows(I1{}, I0{}, KE{});
        __builtin_amdgcn_sched_barrier(0);     // every row load requested so far is in flight before the first one is consumed
        issue_stage(0);
        if (S > 1) issue_stage(1);
        if (S > 2) issue_stage(2);
        __builtin_amdgcn_sched_barrier(0);
        ln_rows(I0{});
        load_rows(I1{}, KE{}, KA{});
        __builtin_amdgcn_sched_barrier(0);
        ln_rows(I1{});
    }

    // ---- main loop over the W stream ---------------------------------------------------------------------------------
    f32x4 acc[8][2];
    // store instructions this wave issues per column tile (a row tile entirely past M issues none: exec == 0 is branched over)
    const int st_per_tile = 2 * ((m0 + wid * 32 < M) + (m0 + wid * 32 + 8 < M) + (m0 + wid * 32 + 16 < M) + (m0 + wid * 32 + 24 < M));
    const int sx = rr & 7;
    const int frag_off = rr * 128;
    for (int nt = 0; nt < ntiles; ++nt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kt = 0; kt < KS; ++kt) {
            const int s = nt * KS + kt;
            // stage s has landed for this wave once <= 4 * min(2, S-1-s) VMEM ops are outstanding (see header)
            {
                // stores of the previous tile that are younger than stage s's loads: 4 per row tile this wave really stored
                const int stp = ((kt <= 2) && (nt > 0)) ? st_per_tile : 0;
                const int rem = S - 1 - s;
                // (a partially valid tail wave stores 2, 4 or 6 times: treated as 0 pending = conservative, never unsafe)
                if (rem >= 2) { if (stp == 8) wait_vmcnt<16>(); else wait_vmcnt<8>(); }
                else if (rem == 1) { if (stp == 8) wait_vmcnt<12>(); else wait_vmcnt<4>(); }
                else { if (stp == 8) wait_vmcnt<8>(); else wait_vmcnt<0>(); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // everyone's share of stage s is in LDS; slot (s+3)%4 is free again
            asm volatile("" ::: "memory");
            if (s + 3 < S) issue_stage(s + 3);
            const unsigned char* st = ring + (s & (PN_NST - 1)) * PN_STAGE_BYTES + frag_off;
            {
                // all 16 W fragments of the stage: 8 requested up front, the 8 of the second k-step one per MFMA pair
                // (pinned with sched_group_barrier: LLVM otherwise serialises 2 reads -> wait -> 4 MFMAs)
                const int so0 = (g ^ sx) * 16, so1 = ((4 + g) ^ sx) * 16;
                bf16x8 wf0[8], wf1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wf0[i] = *reinterpret_cast<const bf16x8*>(st + i * 2048 + so0);
#pragma unroll
                for (int i = 0; i < 8; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(st + i * 2048 + so1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], afrag[0][kt * 2], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[i], afrag[1][kt * 2], acc[i][1], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], afrag[0][kt * 2 + 1], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[i], afrag[1][kt * 2 + 1], acc[i][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        }
        // ---- column-tile epilogue straight from registers -----------------------------------------------------------
        // Per row tile j and column quad qd (64 columns = 128 bytes of bf16) lane (r16, g) holds two 16-byte pieces of row
        // r16: A = columns [8g, 8g+8) and B = columns [32+8g, 32+8g+8) of the quad.  Lanes r16 and r16 ^ 8 swap one piece
        // (DPP row_ror:8) so that a store instruction is fed by 8 lanes per row: rows 0-7 get A from lanes r16 < 8 and B
        // from lanes r16 >= 8 in the first store, rows 8-15 the same in the second -> every store instruction writes 8
        // complete 128-byte lines (the 4-lanes-per-row form ran the store path at 2.3 TB/s, profiles/r01_panel_ablation.log).
        int ntr = nt + rot; if (ntr >= ntiles) ntr -= ntiles;
        const int n0 = ntr * PN_BN;
        // lane-derived values of the epilogue are re-derived here from an opaque lane id: kept live across the MFMA stages they
        // were spilled to scratch, and a scratch reload in the epilogue has to wait (in-order VMEM return) for the weight
        // stages in flight
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int rr = lane_e & 15, g = lane_e >> 4;
        const bool lo = rr < 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mrow = m0 + wid * 32 + j * 16;
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                const int nq = n0 + qd * 64;
                float va[8], vb[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    va[r] = acc[4 * qd][j][r] + sbias[nq + 8 * g + r];
                    va[4 + r] = acc[4 * qd + 1][j][r] + sbias[nq + 8 * g + 4 + r];
                    vb[r] = acc[4 * qd + 2][j][r] + sbias[nq + 32 + 8 * g + r];
                    vb[4 + r] = acc[4 * qd + 3][j][r] + sbias[nq + 32 + 8 * g + 4 + r];
                }
                const u32x4 pa = epi.pack8(va), pb = epi.pack8(vb);
                const u32x4 give = lo ? pb : pa;                  // lanes < 8 hand over B, lanes >= 8 hand over A
                const u32x4 got = swap_half_rows(give);            // ... and receive the partner row's A / B
                // first store: rows 0-7   (lo lanes: own A;            hi lanes: partner (row r16-8) B)
                // second store: rows 8-15 (lo lanes: partner (r16+8) A; hi lanes: own B)
                const int col = nq + (lo ? 8 * g : 32 + 8 * g);
                const int r_first = mrow + (rr & 7), r_second = r_first + 8;
                const u32x4 first = lo ? pa : got, second = lo ? got : pb;
                if (r_first < M) epi.template store_piece<E>(r_first, col, first);
                if (r_second < M) epi.template store_piece<E>(r_second, col, second);
            }
        }
    }
}

template <int E, typename Epi>
inline hipError_t launch_ln_panel_gemm(hipStream_t s, const float* x, const float* gamma, const float* beta, float eps,
                                       const bf16_t* W, const float* bias, int M, int N, const Epi& epi) {
    const size_t lds = (size_t)PN_NST * PN_STAGE_BYTES + (size_t)N * sizeof(float);
    auto kern = ln_panel_gemm_kernel<E, Epi>;
    static LdsAttr attr;                // one per template instantiation
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((M + PN_BM - 1) / PN_BM), dim3(256), lds, s, x, gamma, beta, eps, W, bias, M, N, epi);
    return hipGetLastError();
}

}  // namespace pq
