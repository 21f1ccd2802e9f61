// Row N3 (training step), first correct version: fp32 kernels for the forward / backward of the K-permutation loss (decoder and
// encoder) and the optimiser step.
//
// These are deliberately plain — one generic strided GEMM on the VALU, wave-per-row LayerNorm / cross-entropy backward,
// one workgroup per (image, head) attention forward / backward with everything in LDS — because this round's bar for the
// row is gradient parity with the reference (tests/golden/parseq_train.*), not speed: every kernel below has a one-line CPU
// counterpart in oracle/decoder_backward.py, which is itself checked against autograd.  The MFMA versions (bf16 operands,
// TN / NN tile loaders, fused LayerNorm-backward epilogues) replace them once the whole step is parity-green.
//
// All tensors are fp32, row-major.  Kernels that accumulate say so; everything is deterministic (no atomics).
#pragma once
#include "common.h"
#include <type_traits>

namespace pq {

// -------------------------------------------------------------------------------------------------------------------
// C[m][n] (+)= alpha * sum_k A(m, k) * B(k, n) + bias[n] + R[m % rper][n]
// A(m, k) = A[m * sam + k * sak], B(k, n) = B[k * sbk + n * sbn]: one kernel covers X W^T (forward, dX) and dY^T X (dW).
// -------------------------------------------------------------------------------------------------------------------
struct SgemmArgs {
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    const float* bias;                   // [N] or nullptr
    const float* R; long ldr; int rper;  // residual rows (row m reads R[(m % rper) * ldr + n]) or nullptr
    float* C; long ldc;
    int M, N, K;
    float alpha;
    int accumulate;                      // C += ... instead of C = ...
    float* asum;                         // [M] += sum_k A(m, k) (fp32, before any rounding) or nullptr: the bias gradient riding on the dW
                                         // product dY^T X (A = dY^T), which streams dY anyway — mfma_bgemm_kernel only
    const float* gelu_pre;               // [M][ldc] or nullptr: the stored value is multiplied by gelu'(gelu_pre[m][n]) — the GELU backward
                                         // riding on the dX product through fc2 (d hpre = (dY W2) * gelu'(hpre)) — mfma_bgemm_kernel only
    float* gelu_out;                     // [M][ldc] or nullptr: gelu(stored value) is written here as well (fc1: pre-activation AND activation from one
                                         // epilogue) — mfma_bgemm_kernel only
    // bf16 SHADOW operands (round 3; the matrix-core kernels of the bf16-operand mode only): the producer of an activation writes it as
    // bfloat16 — the same round-to-nearest-even the operand loaders applied on the way into LDS, so the products are bit-identical — and
    // the GEMM reads half the bytes with nothing to convert.  a16 / b16: A / B point at bf16_t data (strides in elements).
    int a16, b16;
    bf16_t* c16;                         // [M][ldc] or nullptr: the stored value again, rounded to bf16 (the next product's operand)
    bf16_t* gelu_out16;                  // [M][ldc] or nullptr: gelu(stored value) as bf16 (instead of gelu_out)
    // bf16-ONLY storage (the step's default in the bf16-operand mode, what bf16-mixed autocast keeps of a Linear output): C may be nullptr
    // when c16 is given — the fp32 copy of the result is not written at all — and the GELU backward can read its pre-activation as bf16
    const bf16_t* gelu_pre16;            // [M][ldc] or nullptr: as gelu_pre, from the bf16 copy of the pre-activation
};
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float gelu_grad(float v);

constexpr int SG_BM = 64, SG_BN = 64, SG_BK = 16;

static __global__ __launch_bounds__(256)
void sgemm_kernel(const SgemmArgs a) {
    __shared__ float As[SG_BK][SG_BM + 4];
    __shared__ float Bs[SG_BK][SG_BN + 4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * SG_BM, n0 = blockIdx.x * SG_BN;
    const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;
    const bool a_kfast = a.sak == 1, b_nfast = a.sbn == 1;      // walk the contiguous axis with consecutive threads
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += SG_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int am = a_kfast ? idx / SG_BK : idx % SG_BM, ak = a_kfast ? idx % SG_BK : idx / SG_BM;
            const int gm = m0 + am, gk = k0 + ak;
            As[ak][am] = (gm < a.M && gk < a.K) ? a.A[(size_t)gm * a.sam + (size_t)gk * a.sak] : 0.f;
            const int bn = b_nfast ? idx % SG_BN : idx / SG_BK, bk = b_nfast ? idx / SG_BN : idx % SG_BK;
            const int gn = n0 + bn, gk2 = k0 + bk;
            Bs[bk][bn] = (gn < a.N && gk2 < a.K) ? a.B[(size_t)gk2 * a.sbk + (size_t)gn * a.sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SG_BK; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[kk][tm + i]; bv[i] = Bs[kk][tn + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + tm + i;
        if (gm >= a.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tn + j;
            if (gn >= a.N) continue;
            float v = a.alpha * acc[i][j];
            if (a.bias) v += a.bias[gn];
            if (a.R) v += a.R[(size_t)(gm % a.rper) * a.ldr + gn];
            float* c = a.C + (size_t)gm * a.ldc + gn;
            *c = a.accumulate ? *c + v : v;
        }
    }
}

// The same contraction on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate) for the shapes that
// carry the step's FLOPs: M % 128 == 0, N % 128 == 0, K % 16 == 0, 16-byte aligned rows.  128 x 128 block tile, four waves of
// 64 x 64, 16 of K per LDS stage, operands parked k-major in LDS (As[k][m], Bs[k][n]) so that a lane's MFMA operand is one
// ds_read_b32 and either memory orientation of A / B (k- or m/n-contiguous) is loaded with 16-byte accesses.
// blockIdx.z splits K (the dW contractions run over batch * tokens rows but have few output tiles): split z handles
// [z * k_chunk, +k_chunk) and, when gridDim.z > 1, writes its tile to partial[z][M][N]; splitk_reduce_kernel folds them in a
// fixed order (deterministic).  bias / residual / accumulate are applied by whichever kernel writes C.
constexpr int MG_BM = 128, MG_BN = 128, MG_BK = 16, MG_LD = 128 + 16;

__device__ __forceinline__ void mg_load_tile(const float* __restrict__ base, long s_outer, long s_k, int outer0, int k0, bool k_fast,
                                            float (*tile)[MG_LD], int tid) {
    // tile[k][o] = base[(outer0 + o) * s_outer + (k0 + k) * s_k]   for o < 128, k < 16; one of s_outer / s_k is 1
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (k_fast) {
            const int o = tid & 127, k4 = (tid >> 7) + 2 * it;             // 4 consecutive k of one row
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(outer0 + o) * s_outer + k0 + 4 * k4);
            tile[4 * k4 + 0][o] = v.x; tile[4 * k4 + 1][o] = v.y; tile[4 * k4 + 2][o] = v.z; tile[4 * k4 + 3][o] = v.w;
        } else {
            const int o4 = tid & 31, k = (tid >> 5) + 8 * it;              // 4 consecutive rows of one k
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(k0 + k) * s_k + outer0 + 4 * o4);
            *reinterpret_cast<float4*>(&tile[k][4 * o4]) = v;
        }
    }
}

static __global__ __launch_bounds__(256)
void mfma_sgemm_kernel(const SgemmArgs a, int k_chunk, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As[MG_BK][MG_LD];
    __shared__ __attribute__((aligned(16))) float Bs[MG_BK][MG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * MG_BM, n0 = blockIdx.x * MG_BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r16 = lane & 15, g = lane >> 4;
    const bool a_kfast = a.sak == 1, b_kfast = a.sbk == 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kbeg = blockIdx.z * k_chunk, kend = min(a.K, kbeg + k_chunk);
    for (int k0 = kbeg; k0 < kend; k0 += MG_BK) {
        mg_load_tile(a.A, a.sam, a.sak, m0, k0, a_kfast, As, tid);
        mg_load_tile(a.B, a.sbn, a.sbk, n0, k0, b_kfast, Bs, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < MG_BK / 4; ++ks) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[4 * ks + g][wm + 16 * i + r16]; bv[i] = Bs[4 * ks + g][wn + 16 * i + r16]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // lane holds D[row = 16 i + 4 g + r][col = 16 j + r16]
    const bool direct = gridDim.z == 1;
    float* out = direct ? a.C : partial + (size_t)blockIdx.z * a.M * a.N;
    const long ldo = direct ? a.ldc : a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = m0 + wm + 16 * i + 4 * g + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gn = n0 + wn + 16 * j + r16;
                float v = acc[i][j][r];
                float* c = out + (size_t)gm * ldo + gn;
                if (direct) {
                    v *= a.alpha;
                    if (a.bias) v += a.bias[gn];
                    if (a.R) v += a.R[(size_t)(gm % a.rper) * a.ldr + gn];
                    if (a.accumulate) v += *c;
                }
                *c = v;
            }
        }
}

// The same contraction with bf16 OPERANDS (every element of A and B rounded to bfloat16, round-to-nearest-even, on its way into LDS),
// fp32 products and accumulation on v_mfma_f32_16x16x32_bf16, fp32 master data in memory: the training step's "bf16" mode
// (BASELINE configs[4] trains bf16-mixed; the gate is oracle.decoder_backward.rounding('bf16'): cosine >= 0.9996 per gradient tensor).
// K % 32 == 0; M and N arbitrary (edge tiles: see bg_fetch).  128 x 128 block tile, four waves of 64 x 64, 32 of K per stage; operands sit in LDS as
// [outer][k] bf16 rows of 80 bytes (64 + 16 pad: the 16 lanes of a ds_read_b128 group fall on distinct banks), so a lane's MFMA operand
// is one ds_read_b128 whichever way the matrix lies in memory; two LDS stages, the next stage's global loads are issued before the
// current stage's MFMAs and converted / stored after them (one barrier per stage).  Split-K and epilogue exactly as mfma_sgemm_kernel.
constexpr int BG_BK = 32, BG_LD = 40;       // elements

// Tiles may hang over the edge of the matrix: over-the-edge lanes re-read the last valid row (k-contiguous operand) or the last valid
// group of four (outer-contiguous operand; the outer extent is a multiple of 4 there) and their products land in accumulator rows /
// columns the epilogue does not store.
// Round 3 (profiles/r03_train_gemm_counters_v0.md: 60 % of the wave cycles parked at s_waitcnt, 10 VALU instructions per MFMA, L2 hit
// rate 42 % with 1.4-3x the algorithmic bytes fetched from HBM):
//   * the kernel is a template on the two operand orientations and every thread's four source pointers per operand are computed ONCE
//     — the loop body used to redo 64-bit index arithmetic with clamps per load behind a run-time orientation branch, and the
//     register shuffling that came with it made the compiler wait for five of the eight loads of stage k + 1 BEFORE the MFMAs of stage
//     k (the disassembly showed `s_waitcnt vmcnt(7)`, `vmcnt(3)` ahead of the first MFMA): the prefetch hid nothing;
//   * XCD-aware tile order: the hardware deals consecutive workgroup ids to the eight XCDs (private L2s) round-robin, so the N-tiles
//     that share an A row panel all landed on different XCDs and each fetched the panel from HBM for itself.  Workgroup id L now maps to
//     logical id (L % 8) * (total / 8) + L / 8 and logical ids walk the N-tiles of one row panel first: a panel's consumers share an L2.
template <bool KFAST, typename T = float>
struct BgOperand {
    static constexpr bool HALF = sizeof(T) == 2;                 // bf16 shadow operand: loaded and parked as it is
    static constexpr int NL = (KFAST && HALF) ? 2 : 4;           // loads per thread and stage
    using V = std::conditional_t<!HALF, float4, std::conditional_t<KFAST, u32x4, u32x2>>;      // native vectors: HIP's uint4 / uint2 structs in an array end up in scratch
    const T* p[NL];               // this thread's loads of the current stage
    long step;                    // pointer increment per 32-deep stage
    __device__ __forceinline__ void init(const T* __restrict__ base, long s_outer, long s_k, int outer0, int limit, int k0, int tid) {
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            if constexpr (KFAST && !HALF) {         // rows of 32 consecutive k: thread = (row idx >> 3, 4 consecutive k at 4 (idx & 7))
                const int idx = tid + 256 * it, o = min(outer0 + (idx >> 3), limit - 1);
                p[it] = base + (size_t)o * s_outer + k0 + 4 * (idx & 7);
            } else if constexpr (KFAST) {           // bf16 rows of 32 k = 64 bytes: thread = (row idx >> 2, 8 consecutive k at 8 (idx & 3))
                const int idx = tid + 256 * it, o = min(outer0 + (idx >> 2), limit - 1);
                p[it] = base + (size_t)o * s_outer + k0 + 8 * (idx & 3);
            } else {                                 // four CONSECUTIVE k of the same four outer indices per thread
                const int k = 4 * (tid >> 5) + it, o = min(outer0 + 4 * (tid & 31), limit - 4);
                p[it] = base + (size_t)(k0 + k) * s_k + o;
            }
        }
        step = KFAST ? (long)BG_BK : (long)BG_BK * s_k;
    }
    __device__ __forceinline__ void fetch(V (&r)[NL]) {
#pragma unroll
        for (int it = 0; it < NL; ++it) { r[it] = *reinterpret_cast<const V*>(p[it]); p[it] += step; }
    }
    static __device__ __forceinline__ void park(bf16_t (*tile)[BG_LD], const V (&r)[NL], int tid) {
        if constexpr (KFAST && !HALF) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + 256 * it, o = idx >> 3, k4 = idx & 7;
                union { uint2 u; bf16_t e[4]; } h;
                h.e[0] = static_cast<bf16_t>(r[it].x); h.e[1] = static_cast<bf16_t>(r[it].y); h.e[2] = static_cast<bf16_t>(r[it].z); h.e[3] = static_cast<bf16_t>(r[it].w);
                *reinterpret_cast<uint2*>(&tile[o][4 * k4]) = h.u;
            }
        } else if constexpr (KFAST) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + 256 * it;
                *reinterpret_cast<u32x4*>(&tile[idx >> 2][8 * (idx & 3)]) = r[it];
            }
        } else if constexpr (!HALF) {
            // r[it] = four outer indices (4 o4 .. 4 o4 + 3) at k = 4 kq + it: transposed in registers, one 8-byte store per outer index
            // (these stores conflict 8-way in LDS — rows four apart are 16 banks apart; rotating each lane's row order made them 2-way and
            // changed nothing measurable: the dW products are not bound by it)
            const int kq = tid >> 5, o4 = tid & 31;
            const float v[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w}, {r[2].x, r[2].y, r[2].z, r[2].w}, {r[3].x, r[3].y, r[3].z, r[3].w}};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                union { uint2 u; bf16_t e[4]; } h;
#pragma unroll
                for (int it = 0; it < 4; ++it) h.e[it] = static_cast<bf16_t>(v[it][i]);
                *reinterpret_cast<uint2*>(&tile[4 * o4 + i][4 * kq]) = h.u;
            }
        } else {
            // the same 4 x 4 transposition on 16-bit values: r[it] = {lo 16 bits of .x: outer 0, hi: outer 1, .y: outer 2, 3} at k = 4 kq + it
            const int kq = tid >> 5, o4 = tid & 31;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned e[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) { const unsigned w = (i & 2) ? r[it].y : r[it].x; e[it] = (i & 1) ? (w >> 16) : (w & 0xffffu); }
                *reinterpret_cast<u32x2*>(&tile[4 * o4 + i][4 * kq]) = u32x2{e[0] | (e[1] << 16), e[2] | (e[3] << 16)};
            }
        }
    }
};

// Epilogue of the bf16-operand kernels.  Lane holds D[row = 16 i + 4 g + r][col = 16 j + r16] of its wave's 64 x 64 quarter.
//
// Staged form (round 3, the usual case): the accumulators go through LDS — the operand tiles are dead — 64 rows at a time, and come back
// as 16-byte row pieces: 32 lanes write 512 contiguous bytes of a row of C (and read the residual / old-C / GELU rows the same way),
// the bf16 shadows leave as 8-byte pieces.  The direct form below it (one 4-byte access per lane and element, 16 lanes = 64 bytes per row
// and instruction, 64 store instructions per lane and output) ran the products whose output is large at 1-2 TB/s of C — the stores, not
// the operand traffic, were what bounded them (49 152 x 1536: 302 MB of C in 169 us; with a second, 2-byte shadow store per element
// 229 us).  Same arithmetic in the same order: (alpha acc + ((bias + residual) + old C)) * gelu'(pre).
// The direct form remains for outputs that are not 16-byte addressable (the 95-class head) and for edge tiles in N.
constexpr int EP_LD = 132;       // floats per staged row: 128 + 4 (the four row groups of a store land on distinct banks)
constexpr int EP_STAGE_BYTES = 64 * EP_LD * 4;
__device__ __forceinline__ bool ep_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// XCD-aware order of the workgroups of a launch.  The hardware deals consecutive workgroup ids (x fastest, then z) to the eight XCDs
// round-robin; id L is mapped to the logical id (L % 8) * (T / 8) + L / 8 so that ONE XCD walks a contiguous range of logical ids, and logical
// ids walk the N-tiles of a row panel first, then the row panels (a panel's consumers share an L2).  The tail that does not fill a group of
// eight keeps its id.
// SPLIT_AWARE (round 3, second half; the fp32-operand kernel): the split index is part of the walk — logical ids walk the tiles of split 0, then
// of split 1, ... — because the tiles of one split all read the same k range of both operands and, spread over eight XCDs, every private L2
// fetched that range for itself.  Counters per dW launch, before -> after: fp32 operands (the decoder's) 673 -> 290 MB from HBM (184 MB of
// operands) and 124 -> 106 us; all-bf16 (the encoder's, mfma_bgemm16t_kernel) 472 -> 222 MB (190 MB of operands) but 102 -> 109 us — that kernel
// was not waiting on HBM, and eight XCDs each serving an eighth of every split's tiles spread its L2 reads better — so the all-bf16 kernels
// keep the split in blockIdx.z (profiles/r03_train_pmc_hbm_traffic.md, r03_train_pmc_fetch_after.md).
struct BgTile { int tm, tn, z; };
template <bool SPLIT_AWARE>
__device__ __forceinline__ BgTile bg_tile(int gn, int gm) {
    const int total = gn * gm;
    const int T = SPLIT_AWARE ? total * (int)gridDim.z : total, L = (int)blockIdx.x + (SPLIT_AWARE ? total * (int)blockIdx.z : 0), whole = T & ~7;
    const int logical = L < whole ? (L & 7) * (whole >> 3) + (L >> 3) : L;
    BgTile t;
    t.z = SPLIT_AWARE ? logical / total : (int)blockIdx.z;
    const int in_split = SPLIT_AWARE ? logical - t.z * total : logical;
    t.tn = in_split % gn; t.tm = in_split / gn;
    return t;
}

__device__ __forceinline__ void bg_epilogue(const SgemmArgs& a, const f32x4 (&acc)[4][4], float* __restrict__ partial, float* __restrict__ stage,
                                            int m0, int n0, int tid, int zsplit) {
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const bool direct = gridDim.z == 1;
    float* __restrict__ out = direct ? a.C : partial + (size_t)zsplit * a.M * a.N;
    const long ldo = direct ? a.ldc : a.N;
    const float* __restrict__ Rb = direct ? a.R : nullptr;
    const bool acc_c = direct && a.accumulate;
    const float alpha = direct ? a.alpha : 1.f;
    bool vec = n0 + MG_BN <= a.N && ldo % 4 == 0 && ep_al16(out);
    if (direct)
        vec = vec && ep_al16(a.gelu_pre16) && ep_al16(a.bias) && ep_al16(a.R) && a.ldr % 4 == 0 && ep_al16(a.gelu_pre) && ep_al16(a.gelu_out) && ep_al16(a.c16) && ep_al16(a.gelu_out16);
    if (vec) {
        const int c4 = tid & 31, gn = n0 + 4 * c4;
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (direct && a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + gn);
        const float* __restrict__ pre = direct ? a.gelu_pre : nullptr;
        const bf16_t* __restrict__ pre16 = direct ? a.gelu_pre16 : nullptr;
        const bool store32 = out != nullptr;
        float* __restrict__ gout = direct ? a.gelu_out : nullptr;
        bf16_t* __restrict__ c16 = direct ? a.c16 : nullptr;
        bf16_t* __restrict__ g16 = direct ? a.gelu_out16 : nullptr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((wave >> 1) == h) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) stage[(16 * i + 4 * g + r) * EP_LD + wn + 16 * j + r16] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                // two row pieces per thread and round (four would spill at three waves per SIMD): everything they read is requested
                // before the first of them is stored
                constexpr int NQ = 2;
                f32x4 v[NQ], rv[NQ], cv[NQ], pv[NQ];
                bool ok[NQ];
                size_t at[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int row = (tid >> 5) + 8 * (NQ * qq + q), gm_ = m0 + 64 * h + row;
                    ok[q] = gm_ < a.M;
                    at[q] = (size_t)gm_ * ldo + gn;
                    v[q] = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + 4 * c4);
                    rv[q] = cv[q] = pv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (Rb && ok[q]) rv[q] = *reinterpret_cast<const f32x4*>(Rb + (size_t)(gm_ < a.rper ? gm_ : gm_ % a.rper) * a.ldr + gn);
                    if (acc_c && ok[q]) cv[q] = *reinterpret_cast<const f32x4*>(out + at[q]);
                    if (pre && ok[q]) pv[q] = *reinterpret_cast<const f32x4*>(pre + at[q]);
                    if (pre16 && ok[q]) {
                        const u32x2 w = *reinterpret_cast<const u32x2*>(pre16 + at[q]);
                        pv[q] = f32x4{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)};
                    }
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (!ok[q]) continue;
                    f32x4 add = b4;
                    if (Rb) add += rv[q];
                    if (acc_c) add += cv[q];
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf(alpha, v[q][e], add[e]);      // spelled out: both forms and every instantiation round alike
                    if (pre || pre16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] *= gelu_grad(pv[q][e]);
                    }
                    if (store32) *reinterpret_cast<f32x4*>(out + at[q]) = o;
                    if (c16) {
                        union { u32x2 u; bf16_t e[4]; } hh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hh.e[e] = static_cast<bf16_t>(o[e]);
                        *reinterpret_cast<u32x2*>(c16 + at[q]) = hh.u;
                    }
                    if (gout || g16) {
                        f32x4 ge;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ge[e] = gelu_erf(o[e]);
                        if (gout) *reinterpret_cast<f32x4*>(gout + at[q]) = ge;
                        if (g16) {
                            union { u32x2 u; bf16_t e[4]; } hh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) hh.e[e] = static_cast<bf16_t>(ge[e]);
                            *reinterpret_cast<u32x2*>(g16 + at[q]) = hh.u;
                        }
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    const int gn0 = n0 + wn + r16;
    bool cok[4];
    float bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cok[j] = gn0 + 16 * j < a.N;
        bj[j] = (direct && a.bias && cok[j]) ? a.bias[gn0 + 16 * j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm_ = m0 + wm + 16 * i + 4 * g + r;
            if (gm_ >= a.M) continue;
            float* __restrict__ crow = out + (size_t)gm_ * ldo + gn0;
            float add[4] = {bj[0], bj[1], bj[2], bj[3]};
            if (Rb) {
                const float* __restrict__ rrow = Rb + (size_t)(gm_ < a.rper ? gm_ : gm_ % a.rper) * a.ldr + gn0;
                float rv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[j] = cok[j] ? rrow[16 * j] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) add[j] += rv[j];
            }
            if (acc_c) {
                float cv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cv[j] = cok[j] ? crow[16 * j] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) add[j] += cv[j];
            }
            float mul[4] = {1.f, 1.f, 1.f, 1.f};
            if (direct && a.gelu_pre16) {
                const bf16_t* __restrict__ prow = a.gelu_pre16 + (size_t)gm_ * a.ldc + gn0;
#pragma unroll
                for (int j = 0; j < 4; ++j) mul[j] = gelu_grad(cok[j] ? static_cast<float>(prow[16 * j]) : 0.f);
            }
            if (direct && a.gelu_pre) {
                const float* __restrict__ prow = a.gelu_pre + (size_t)gm_ * a.ldc + gn0;
                float pv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pv[j] = cok[j] ? prow[16 * j] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) mul[j] = gelu_grad(pv[j]);
            }
            float* __restrict__ grow = (direct && a.gelu_out) ? a.gelu_out + (size_t)gm_ * a.ldc + gn0 : nullptr;
            bf16_t* __restrict__ c16row = (direct && a.c16) ? a.c16 + (size_t)gm_ * a.ldc + gn0 : nullptr;
            bf16_t* __restrict__ g16row = (direct && a.gelu_out16) ? a.gelu_out16 + (size_t)gm_ * a.ldc + gn0 : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (cok[j]) {
                    const float v = fmaf(alpha, acc[i][j][r], add[j]) * mul[j];
                    if (out) crow[16 * j] = v;
                    if (c16row) c16row[16 * j] = static_cast<bf16_t>(v);
                    if (grow) grow[16 * j] = gelu_erf(v);
                    if (g16row) g16row[16 * j] = static_cast<bf16_t>(gelu_erf(v));
                }
        }
}

// grid: (tiles_n * tiles_m, 1, splits) workgroups; gn, gm = the tile counts.  B16: the B operand is a bf16 shadow (a.b16).
// A16: so is the A operand (a.a16; outer-contiguous only: the dW products' dY^T, whose row sums — the bias gradient — are then the sums
// of the bf16 values)
template <bool AKF, bool BKF, bool B16 = false, bool A16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void mfma_bgemm_kernel(const SgemmArgs a, int k_chunk, float* __restrict__ partial, int gn, int gm) {
    static_assert(!(A16 && AKF), "a k-contiguous bf16 A goes to mfma_bgemm16_kernel");
    using TB = std::conditional_t<B16, bf16_t, float>;
    using OpB = BgOperand<BKF, TB>;
    using TA = std::conditional_t<A16, bf16_t, float>;
    using OpA = BgOperand<AKF, TA>;
    // one LDS block: the two operands' two stages, then (all of it) the epilogue's staging tile
    constexpr int TILE_BYTES = 2 * MG_BM * BG_LD * 2;
    static_assert(2 * TILE_BYTES >= EP_STAGE_BYTES && 2 * TILE_BYTES >= 8 * 128 * 4, "LDS block too small for the epilogue");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    bf16_t (*As)[MG_BM][BG_LD] = reinterpret_cast<bf16_t (*)[MG_BM][BG_LD]>(smem);
    bf16_t (*Bs)[MG_BN][BG_LD] = reinterpret_cast<bf16_t (*)[MG_BN][BG_LD]>(smem + TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (see above); the tail that does not fill a whole group of eight keeps its id
    const BgTile bt = bg_tile<true>(gn, gm);
    const int tn_ = bt.tn, tm_ = bt.tm, zsplit = bt.z;
    const int m0 = tm_ * MG_BM, n0 = tn_ * MG_BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kbeg = zsplit * k_chunk, kend = min(a.K, kbeg + k_chunk);
    OpA oa; OpB ob;
    oa.init(reinterpret_cast<const TA*>(a.A), a.sam, a.sak, m0, a.M, kbeg, tid);
    ob.init(reinterpret_cast<const TB*>(a.B), a.sbn, a.sbk, n0, a.N, kbeg, tid);
    typename OpA::V ra[OpA::NL];
    typename OpB::V rb[OpB::NL];
    // row sums of A over this workgroup's k range (a.asum; only the first N-tile of a row panel adds them up): rs[i] belongs to outer
    // index 4 (tid & 31) + i (outer-contiguous A) or to row (tid >> 3) + 32 i (k-contiguous A: eight lanes per row)
    const bool do_sum = a.asum != nullptr && tn_ == 0;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    auto add_rows = [&]() {
        if constexpr (A16) {
#pragma unroll
            for (int it = 0; it < 4; ++it) { rs[0] += bf16_lo(ra[it].x); rs[1] += bf16_hi(ra[it].x); rs[2] += bf16_lo(ra[it].y); rs[3] += bf16_hi(ra[it].y); }
        } else if constexpr (AKF) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rs[it] += (ra[it].x + ra[it].y) + (ra[it].z + ra[it].w);
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) { rs[0] += ra[it].x; rs[1] += ra[it].y; rs[2] += ra[it].z; rs[3] += ra[it].w; }
        }
    };
    if (kbeg < kend) {
        oa.fetch(ra); ob.fetch(rb);
        if (do_sum) add_rows();
        OpA::park(As[0], ra, tid);
        OpB::park(Bs[0], rb, tid);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BG_BK) {
        const bool more = k0 + BG_BK < kend;
        if (more) { oa.fetch(ra); ob.fetch(rb); }       // in flight under this stage's MFMAs; first touched by park() below
        bf16x8 av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = *reinterpret_cast<const bf16x8*>(&As[cur][wm + 16 * i + r16][8 * g]);
            bv[i] = *reinterpret_cast<const bf16x8*>(&Bs[cur][wn + 16 * i + r16][8 * g]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);              // nothing of park() (its waits for the loads) moves above the MFMAs
        if (more) {
            if (do_sum) add_rows();
            OpA::park(As[cur ^ 1], ra, tid);
            OpB::park(Bs[cur ^ 1], rb, tid);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (do_sum) {
        // fold the threads' partial row sums in a fixed order through LDS (the operand tiles are dead) and add them to a.asum
        // (split-K: to this split's slot behind the product's partials; splitk_reduce_kernel adds the slots up)
        float* red = reinterpret_cast<float*>(smem);                   // [8][128]
        if constexpr (AKF) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float v = rs[it];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                if ((tid & 7) == 0) red[(tid >> 3) + 32 * it] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(tid >> 5) * 128 + 4 * (tid & 31) + i] = rs[i];
        }
        __syncthreads();
        if (tid < 128 && m0 + tid < a.M) {
            float v;
            if constexpr (AKF) v = red[tid];
            else v = ((red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid])) + ((red[512 + tid] + red[640 + tid]) + (red[768 + tid] + red[896 + tid]));
            if (gridDim.z == 1) a.asum[m0 + tid] += v;
            else partial[(size_t)gridDim.z * a.M * a.N + (size_t)zsplit * a.M + m0 + tid] = v;
        }
        __syncthreads();
    }
    bg_epilogue(a, acc, partial, reinterpret_cast<float*>(smem), m0, n0, tid, zsplit);
}

// Both operands bf16 shadows with k contiguous (the forward products x W^T of the encoder, and dX = dY W through the transposed
// weight shadow): 64 of K per stage — a row of a stage is 128 bytes, one whole cache line per row and request, where a 32-deep stage of
// bf16 would use half of every line it pulls into the CU's L1 — loaded as 16-byte pieces and parked in LDS as they are (no conversion,
// no VALU work between the load and the ds_write_b128).  One LDS buffer of 128 x (64 + 8) per operand (36 KiB: three workgroups per CU
// as before), the next stage waits in registers under the current stage's 32 MFMAs per wave.  Same tile order, accumulation order
// (ascending k in steps of 32), split-K and epilogue as mfma_bgemm_kernel: bit-identical results.  K % 64 == 0.
constexpr int BH_BK = 64, BH_LD = 72;
// one 32-deep half of a 64-deep stage: 16 MFMAs of a wave's 64 x 64 tile.  ONE_B: the B fragments one at a time (the four-workgroup forms)
template <bool ONE_B>
__device__ __forceinline__ void bg16_stage_mfma(const bf16_t (*As)[72], const bf16_t (*Bs)[72], f32x4 (&acc)[4][4], int wm, int wn, int r16, int g, int kk) {
    bf16x8 av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const bf16x8*>(&As[wm + 16 * i + r16][32 * kk + 8 * g]);
    if constexpr (ONE_B) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 bj = *reinterpret_cast<const bf16x8*>(&Bs[wn + 16 * j + r16][32 * kk + 8 * g]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bj, acc[i][j], 0, 0, 0);
        }
    } else {
        bf16x8 bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const bf16x8*>(&Bs[wn + 16 * i + r16][32 * kk + 8 * g]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
}
// WHOLE (M and N multiples of 128 — every product of the PARSeq-S / ViTSTR encoders): FOUR workgroups per CU.  The kernel waits on memory, not on
// the matrix pipe (six 64-deep stages per tile at K = 384, one stage of register prefetch), so what it needs is more waves to switch to; at 140
// registers it sat at three.  The loads become buffer loads — one resource per operand in SGPRs, ONE constant byte offset per thread and
// operand in a VGPR, the piece's 32-row distance and the k position in the scalar offset: no 64-bit pointers, no per-piece offsets — and the
// B fragments are read one at a time (16 instead of 32 fragment registers): 126 VGPRs, no scratch.
template <bool WHOLE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WHOLE ? 4 : 3, WHOLE ? 4 : 3)))
void mfma_bgemm16_kernel(const SgemmArgs a, int k_chunk, float* __restrict__ partial, int gn, int gm) {
    constexpr int TILE_BYTES = MG_BM * BH_LD * 2;
    static_assert(2 * TILE_BYTES >= EP_STAGE_BYTES, "LDS block too small for the epilogue");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    bf16_t (*As)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem);
    bf16_t (*Bs)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem + TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BgTile bt = bg_tile<false>(gn, gm);
    const int tn_ = bt.tn, tm_ = bt.tm, zsplit = bt.z;
    const int m0 = tm_ * MG_BM, n0 = tn_ * MG_BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kbeg = zsplit * k_chunk, kend = min(a.K, kbeg + k_chunk);
    // thread = (row idx >> 3, 8 consecutive k at 8 (idx & 7)), idx = tid + 256 it; rows past the edge re-read the last valid row
    const bf16_t* pa[WHOLE ? 1 : 4];
    const bf16_t* pb[WHOLE ? 1 : 4];
    __amdgpu_buffer_rsrc_t ares, bres;
    unsigned oa = 0, ob = 0, pa_step = 0, pb_step = 0, kbyte = 0;
    if constexpr (WHOLE) {
        ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, 0x7FFFF000, 0x00020000);
        bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, 0x7FFFF000, 0x00020000);
        oa = 2u * ((unsigned)(m0 + (tid >> 3)) * (unsigned)a.sam + 8u * (tid & 7));
        ob = 2u * ((unsigned)(n0 + (tid >> 3)) * (unsigned)a.sbn + 8u * (tid & 7));
        pa_step = 64u * (unsigned)a.sam; pb_step = 64u * (unsigned)a.sbn;      // 32 rows, in bytes
        kbyte = 2u * (unsigned)kbeg;
    } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 256 * it;
            pa[it] = reinterpret_cast<const bf16_t*>(a.A) + (size_t)min(m0 + (idx >> 3), a.M - 1) * a.sam + kbeg + 8 * (idx & 7);
            pb[it] = reinterpret_cast<const bf16_t*>(a.B) + (size_t)min(n0 + (idx >> 3), a.N - 1) * a.sbn + kbeg + 8 * (idx & 7);
        }
    }
    u32x4 ra[4], rb[4];
    auto fetch = [&]() {
        if constexpr (WHOLE) {
#pragma unroll
            for (int it = 0; it < 4; ++it) ra[it] = __builtin_amdgcn_raw_buffer_load_b128(ares, oa, kbyte + it * pa_step, 0);
#pragma unroll
            for (int it = 0; it < 4; ++it) rb[it] = __builtin_amdgcn_raw_buffer_load_b128(bres, ob, kbyte + it * pb_step, 0);
            kbyte += 2u * BH_BK;
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) { ra[it] = *reinterpret_cast<const u32x4*>(pa[it]); pa[it] += BH_BK; }
#pragma unroll
            for (int it = 0; it < 4; ++it) { rb[it] = *reinterpret_cast<const u32x4*>(pb[it]); pb[it] += BH_BK; }
        }
    };
    if (kbeg < kend) fetch();
    for (int k0 = kbeg; k0 < kend; k0 += BH_BK) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 256 * it;
            *reinterpret_cast<u32x4*>(&As[idx >> 3][8 * (idx & 7)]) = ra[it];
            *reinterpret_cast<u32x4*>(&Bs[idx >> 3][8 * (idx & 7)]) = rb[it];
        }
        __syncthreads();
        if (k0 + BH_BK < kend) fetch();                 // in flight under this stage's MFMAs; first touched by the stores above, next round
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bg16_stage_mfma<WHOLE>(As, Bs, acc, wm, wn, r16, g, kk);
        }
        __builtin_amdgcn_sched_barrier(0);              // the waits for the loads stay below the MFMAs
        __syncthreads();
    }
    bg_epilogue(a, acc, partial, reinterpret_cast<float*>(smem), m0, n0, tid, zsplit);
}

// The dW products dY^T X with BOTH operands bf16 in memory and outer-contiguous (the contraction index m is the row index of dY [m, n]
// and of X [m, k']): 64 rows of m per stage.  A thread loads 8-byte pieces (four outer indices at one m), eight of them per operand
// and stage, and parks them transposed: for each of its four outer indices the eight m values as ONE 16-byte LDS store.  One LDS buffer
// + the next stage in registers, as mfma_bgemm16_kernel; with 32-deep stages the kernel paid one exposed memory round trip per 16
// MFMAs per wave (49 152-deep contractions in 15 splits: 102 stages of 1.9 us) — at 64 deep it pays one per 32.  Row sums of A (the bias
// gradient, a.asum) are sums of the bf16 values.  K % 64 == 0, M % 4 == 0, N % 4 == 0.
// WHOLE: as mfma_bgemm16_kernel<true> — four workgroups per CU, buffer loads with the k position of a piece in the scalar offset (128 VGPRs, no scratch)
template <bool WHOLE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WHOLE ? 4 : 3, WHOLE ? 4 : 3)))
void mfma_bgemm16t_kernel(const SgemmArgs a, int k_chunk, float* __restrict__ partial, int gn, int gm) {
    constexpr int TILE_BYTES = MG_BM * BH_LD * 2;
    static_assert(2 * TILE_BYTES >= EP_STAGE_BYTES && 2 * TILE_BYTES >= 8 * 128 * 4, "LDS block too small for the epilogue");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    bf16_t (*As)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem);
    bf16_t (*Bs)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem + TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BgTile bt = bg_tile<false>(gn, gm);
    const int tn_ = bt.tn, tm_ = bt.tm, zsplit = bt.z;
    const int m0 = tm_ * MG_BM, n0 = tn_ * MG_BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kbeg = zsplit * k_chunk, kend = min(a.K, kbeg + k_chunk);
    // thread = (outer group o4 = tid & 31: outer indices 4 o4 .. 4 o4 + 3, k octet kq = tid >> 5: k = 8 kq + it); groups past the edge
    // re-read the last valid group of four
    const int o4 = tid & 31, kq = tid >> 5;
    const bf16_t* pa = nullptr; const bf16_t* pb = nullptr;
    const long sa = a.sak, sb = a.sbk;
    __amdgpu_buffer_rsrc_t ares, bres;
    unsigned sa2 = 0, sb2 = 0, oa = 0, ob = 0, ka = 0, kb = 0;
    if constexpr (WHOLE) {
        ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, 0x7FFFF000, 0x00020000);
        bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, 0x7FFFF000, 0x00020000);
        sa2 = 2u * (unsigned)a.sak; sb2 = 2u * (unsigned)a.sbk;      // bytes per k
        oa = 8u * (unsigned)kq * sa2 + 2u * (unsigned)(m0 + 4 * o4);
        ob = 8u * (unsigned)kq * sb2 + 2u * (unsigned)(n0 + 4 * o4);
        ka = (unsigned)kbeg * sa2; kb = (unsigned)kbeg * sb2;
    } else {
        pa = reinterpret_cast<const bf16_t*>(a.A) + (size_t)(kbeg + 8 * kq) * a.sak + min(m0 + 4 * o4, a.M - 4);
        pb = reinterpret_cast<const bf16_t*>(a.B) + (size_t)(kbeg + 8 * kq) * a.sbk + min(n0 + 4 * o4, a.N - 4);
    }
    u32x2 ra[8], rb[8];
    auto fetch = [&]() {
        if constexpr (WHOLE) {
#pragma unroll
            for (int it = 0; it < 8; ++it) ra[it] = __builtin_amdgcn_raw_buffer_load_b64(ares, oa, ka + it * sa2, 0);
#pragma unroll
            for (int it = 0; it < 8; ++it) rb[it] = __builtin_amdgcn_raw_buffer_load_b64(bres, ob, kb + it * sb2, 0);
            ka += BH_BK * sa2; kb += BH_BK * sb2;
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) ra[it] = *reinterpret_cast<const u32x2*>(pa + it * sa);
#pragma unroll
            for (int it = 0; it < 8; ++it) rb[it] = *reinterpret_cast<const u32x2*>(pb + it * sb);
            pa += BH_BK * sa; pb += BH_BK * sb;
        }
    };
    // r[it] = {outer 0 | outer 1 << 16, outer 2 | outer 3 << 16} at k = 8 kq + it  ->  row (4 o4 + i): k = 8 kq .. 8 kq + 7 as four dwords
    auto park = [&](bf16_t (*tile)[BH_LD], const u32x2 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned w0 = (i & 2) ? r[2 * d].y : r[2 * d].x, w1 = (i & 2) ? r[2 * d + 1].y : r[2 * d + 1].x;
                o[d] = (i & 1) ? ((w0 >> 16) | (w1 & 0xffff0000u)) : ((w0 & 0xffffu) | (w1 << 16));
            }
            *reinterpret_cast<u32x4*>(&tile[4 * o4 + i][8 * kq]) = o;
        }
    };
    const bool do_sum = a.asum != nullptr && tn_ == 0;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (kbeg < kend) fetch();
    for (int k0 = kbeg; k0 < kend; k0 += BH_BK) {
        if (do_sum) {
#pragma unroll
            for (int it = 0; it < 8; ++it) { rs[0] += bf16_lo(ra[it].x); rs[1] += bf16_hi(ra[it].x); rs[2] += bf16_lo(ra[it].y); rs[3] += bf16_hi(ra[it].y); }
        }
        park(As, ra);
        park(Bs, rb);
        __syncthreads();
        if (k0 + BH_BK < kend) fetch();                 // in flight under this stage's MFMAs
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bg16_stage_mfma<WHOLE>(As, Bs, acc, wm, wn, r16, g, kk);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if (do_sum) {
        // as mfma_bgemm_kernel: the eight k octets' partial sums of every row through LDS in a fixed order
        float* red = reinterpret_cast<float*>(smem);                   // [8][128]
#pragma unroll
        for (int i = 0; i < 4; ++i) red[kq * 128 + 4 * o4 + i] = rs[i];
        __syncthreads();
        if (tid < 128 && m0 + tid < a.M) {
            const float v = ((red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid])) + ((red[512 + tid] + red[640 + tid]) + (red[768 + tid] + red[896 + tid]));
            if (gridDim.z == 1) a.asum[m0 + tid] += v;
            else partial[(size_t)gridDim.z * a.M * a.N + (size_t)zsplit * a.M + m0 + tid] = v;
        }
        __syncthreads();
    }
    bg_epilogue(a, acc, partial, reinterpret_cast<float*>(smem), m0, n0, tid, zsplit);
}

// bf16 shadows of a Linear weight W [N, K] (fp32 master): W16 [N, K] and its transpose Wt16 [K, N], once per step.  N, K multiples of 32.
// All of a step's weight shadows in ONE launch: the table lists the matrices (master offset, shape, first tile, destination relative to the
// shadows' base: W16 there, Wt16 right behind it); workgroup t converts tile t - tile0 of the matrix whose range holds t.
struct ShadowEntry { unsigned src, N, K, tile0; unsigned long long dst; };
static __global__ __launch_bounds__(256)
void weight_shadows_kernel(const float* __restrict__ master, const ShadowEntry* __restrict__ tab, int entries, bf16_t* __restrict__ base) {
    __shared__ float t[32][33];
    int e = 0;
    while (e + 1 < entries && tab[e + 1].tile0 <= blockIdx.x) ++e;
    const ShadowEntry se = tab[e];
    const int N = (int)se.N, K = (int)se.K, tile = (int)(blockIdx.x - se.tile0), kt = K / 32;
    const float* W = master + se.src;
    bf16_t* W16 = base + se.dst; bf16_t* Wt16 = W16 + (size_t)N * K;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n0 = (tile / kt) * 32, k0 = (tile % kt) * 32;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const float v = W[(size_t)(n0 + r) * K + k0 + tx];
        t[r][tx] = v;
        W16[(size_t)(n0 + r) * K + k0 + tx] = static_cast<bf16_t>(v);
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) Wt16[(size_t)(k0 + r) * N + n0 + tx] = static_cast<bf16_t>(t[tx][r]);
}

// dst[Rd, Cd] = src[Rs, Cs] in its top-left corner, zeros elsewhere (Rd >= Rs, Cd >= Cs)
static __global__ __launch_bounds__(256)
void pad_copy_kernel(const float* __restrict__ src, int Rs, int Cs, float* __restrict__ dst, int Rd, int Cd) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Rd * Cd) return;
    const int r = (int)(i / Cd), c = (int)(i % Cd);
    dst[i] = (r < Rs && c < Cs) ? src[(size_t)r * Cs + c] : 0.f;
}
static __global__ __launch_bounds__(256)
void add_into_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

static __global__ __launch_bounds__(256)
void splitk_reduce_kernel(const SgemmArgs a, const float* __restrict__ partial, int splits) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)a.M * a.N;
    if (idx >= total) {                       // the threads past the product fold the row sums of A (partials behind the product's, [splits][M])
        const size_t m = idx - total;
        if (a.asum && m < (size_t)a.M) {
            float t = 0.f;
            for (int z = 0; z < splits; ++z) t += partial[(size_t)splits * total + (size_t)z * a.M + m];
            a.asum[m] += t;
        }
        return;
    }
    const int gm = (int)(idx / a.N), gn = (int)(idx % a.N);
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += partial[(size_t)z * total + idx];
    v *= a.alpha;
    if (a.bias) v += a.bias[gn];
    if (a.R) v += a.R[(size_t)(gm % a.rper) * a.ldr + gn];
    float* c = a.C + (size_t)gm * a.ldc + gn;
    if (a.accumulate) v += *c;
    if (a.gelu_pre) v *= gelu_grad(a.gelu_pre[(size_t)gm * a.ldc + gn]);
    if (a.gelu_pre16) v *= gelu_grad(static_cast<float>(a.gelu_pre16[(size_t)gm * a.ldc + gn]));
    if (a.C) *c = v;
    if (a.c16) a.c16[(size_t)gm * a.ldc + gn] = static_cast<bf16_t>(v);
    if (a.gelu_out) a.gelu_out[(size_t)gm * a.ldc + gn] = gelu_erf(v);
    if (a.gelu_out16) a.gelu_out16[(size_t)gm * a.ldc + gn] = static_cast<bf16_t>(gelu_erf(v));
}

// out[n] (+)= sum_m A[m * lda + n]: bias gradients, LayerNorm affine gradients, sums over the batch ([B, L * E] views).
// Many rows: two deterministic stages (row chunks -> partials, then this kernel again over the partials).
// 16 row groups x 64 columns per workgroup, rows added in a fixed order.  blockIdx.y = row chunk of `rows_per` rows; with more
// than one chunk the kernel writes out[chunk][n] (a partial, never accumulated into).
static __global__ __launch_bounds__(1024)
void colsum_kernel(const float* __restrict__ A, long lda, int M, int N, float* __restrict__ out, int accumulate, int rows_per,
                   float* __restrict__ out2 = nullptr, int split = 0) {      // out2 (single-chunk form only): columns >= split go to out2[n - split]
    __shared__ float part[16][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int m_lo = blockIdx.y * rows_per, m_hi = min(M, m_lo + rows_per);
    float s = 0.f;
    if (n < N)
        for (int m = m_lo + rg; m < m_hi; m += 16) s += A[(size_t)m * lda + n];
    part[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][c];
        float* o = (out2 && n >= split) ? out2 + (n - split) : out + (size_t)blockIdx.y * N + n;
        *o = (accumulate && gridDim.y == 1) ? *o + t : t;
    }
}

// LayerNorm forward of the training encoder, E <= 768 and even; one row per wave, a lane owns column pairs (2 lane + 128 i).  Two-pass
// statistics like rowops.h layernorm_kernel; the affine step is ONE explicit fma, so that the fp32 output and the bf16 output (the
// operand shadow of the products that follow, two elements per 4-byte store) are roundings of the same value whatever the compiler
// does with each instantiation (layernorm_kernel<float> and <bf16> contract it differently: 5 of 1.5 M elements a bf16 ulp apart).
template <typename TO>
__global__ __launch_bounds__(256)
void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, TO* __restrict__ out, int rows, int E, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* __restrict__ xr = x + (size_t)row * E;
    f32x2 v[6];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = 2 * lane + 128 * i;
        v[i] = c < E ? *reinterpret_cast<const f32x2*>(xr + c) : f32x2{0.f, 0.f};
        s += v[i][0] + v[i][1];
    }
    const float inv = 1.0f / (float)E;
    const float mean = wave_sum(s) * inv;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (2 * lane + 128 * i < E) { const float d0 = v[i][0] - mean, d1 = v[i][1] - mean; ss += d0 * d0 + d1 * d1; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * inv + eps);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = 2 * lane + 128 * i;
        if (c < E) {
            const f32x2 wv = *reinterpret_cast<const f32x2*>(w + c), bv = *reinterpret_cast<const f32x2*>(b + c);
            const float y0 = fmaf((v[i][0] - mean) * rstd, wv[0], bv[0]), y1 = fmaf((v[i][1] - mean) * rstd, wv[1], bv[1]);
            if constexpr (sizeof(TO) == 2) {
                union { unsigned u; bf16_t e[2]; } h;
                h.e[0] = static_cast<bf16_t>(y0); h.e[1] = static_cast<bf16_t>(y1);
                *reinterpret_cast<unsigned*>(out + (size_t)row * E + c) = h.u;
            } else {
                *reinterpret_cast<f32x2*>(out + (size_t)row * E + c) = f32x2{y0, y1};
            }
        }
    }
}

// LayerNorm backward, statistics recomputed from x (E <= 768); a workgroup owns LNB_ROWS consecutive rows, one wave per row at a time:
//   dx_out = (add ? add : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w
//   partial[chunk][0 .. E)   = sum over the chunk's rows of dy * xhat   (weight gradient)
//   partial[chunk][E .. 2E)  = sum over the chunk's rows of dy          (bias gradient)
// summed per lane over the wave's rows in ascending order, then over the four waves in wave order: deterministic.  The host folds the
// chunks with colsum_kernel.  (Round 3: the first form wrote dy * xhat as a [rows, E] matrix and ran two column-sum passes over it and
// over dy — 225 MB of extra traffic and two more launches per LayerNorm at 49 152 rows.)
constexpr int LNB_ROWS = 4;       // one row per wave: 64 and 32 rows per workgroup (a serial row loop per wave, even with the next row prefetched) ran the kernel at 68-75 us where one row per wave runs it at HBM speed
static __global__ __launch_bounds__(256)
void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy, const float* __restrict__ add,
                   float* __restrict__ dx_out, float* __restrict__ partial, int rows, int E, float eps, bf16_t* __restrict__ dx16) {
    __shared__ float red[4][2][768];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pg[12], pb[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
    const float inv = 1.0f / (float)E;
    const int r_first = blockIdx.x * LNB_ROWS + wave * (LNB_ROWS / 4);
    // the next row's x and dy are requested before the current row is worked on (the row loop is a chain of wave reductions: without
    // the prefetch every row paid its own memory round trip)
    float nx[12], nd[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < E && r_first < rows;
        nx[i] = ok ? x[(size_t)r_first * E + c] : 0.f;
        nd[i] = ok ? dy[(size_t)r_first * E + c] : 0.f;
    }
    for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
        const int r = r_first + rr;
        if (r >= rows) break;
        const size_t base = (size_t)r * E;
        float xv[12], gv[12], dv[12];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) { xv[i] = nx[i]; dv[i] = nd[i]; s += xv[i]; }
        if (rr + 1 < LNB_ROWS / 4 && r + 1 < rows) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int c = lane + 64 * i;
                nx[i] = c < E ? x[base + E + c] : 0.f;
                nd[i] = c < E ? dy[base + E + c] : 0.f;
            }
        }
        const float mean = wave_sum(s) * inv;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = lane + 64 * i;
            const float d = c < E ? xv[i] - mean : 0.f;
            ss += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) * inv + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = lane + 64 * i;
            xv[i] = c < E ? (xv[i] - mean) * rstd : 0.f;          // xhat
            gv[i] = c < E ? dv[i] * w[c] : 0.f;
            s1 += gv[i];
            s2 += gv[i] * xv[i];
        }
        const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = lane + 64 * i;
            if (c < E) {
                float d = rstd * (gv[i] - m1 - xv[i] * m2);
                if (add) d += add[base + c];
                dx_out[base + c] = d;
                if (dx16) dx16[base + c] = static_cast<bf16_t>(d);      // the shadow the next dX product reads
                pg[i] += dv[i] * xv[i];
                pb[i] += dv[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int c = lane + 64 * i;
        if (c < E) { red[wave][0][c] = pg[i]; red[wave][1][c] = pb[i]; }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * 2 * E;
    for (int c = threadIdx.x; c < 2 * E; c += 256) {
        const int which = c >= E, cc = which ? c - E : c;
        out[c] = ((red[0][which][cc] + red[1][which][cc]) + red[2][which][cc]) + red[3][which][cc];
    }
}

// exact-erf GELU and its derivative (F.gelu default; modules.py:43,77)
static __global__ __launch_bounds__(256)
void gelu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ act, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;      // four elements per thread (16-byte accesses); the tail one by one
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(pre + i);
        *reinterpret_cast<float4*>(act + i) = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
    } else {
        for (size_t j = i; j < n; ++j) act[j] = gelu_erf(pre[j]);
    }
}
__device__ __forceinline__ float gelu_grad(float v) {
    const float cdf = 0.5f * (1.0f + fast_erf(v * 0.70710678118654752440f));
    const float pdf = __expf(-0.5f * v * v) * 0.39894228040143267794f;
    return fmaf(v, pdf, cdf);
}
static __global__ __launch_bounds__(256)
void gelu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dact, float* __restrict__ dpre, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(pre + i), d = *reinterpret_cast<const float4*>(dact + i);
        *reinterpret_cast<float4*>(dpre + i) = make_float4(d.x * gelu_grad(v.x), d.y * gelu_grad(v.y), d.z * gelu_grad(v.z), d.w * gelu_grad(v.w));
    } else {
        for (size_t j = i; j < n; ++j) dpre[j] = dact[j] * gelu_grad(pre[j]);
    }
}

// dst[i] = src[i] over a table of pieces, one workgroup per piece (parseq_model_get_params: the master weights back into the caller's tensors)
struct CopyPiece { const float* src; float* dst; int n; };
constexpr int COPY_PIECE_ELEMS = 8192;
static __global__ __launch_bounds__(256)
void copy_pieces_kernel(const CopyPiece* __restrict__ pieces) {
    const CopyPiece c = pieces[blockIdx.x];
    for (int i = threadIdx.x; i < c.n; i += 256) c.dst[i] = c.src[i];
}

// y = a + b (elementwise)
static __global__ __launch_bounds__(256)
void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}

// Content stream of the teacher-forced decode (model.py:95-98): row (b, j) = sqrt(E) * emb[tok[b][j]] + (j ? pos_queries[j-1] : 0)
static __global__ __launch_bounds__(256)
void train_content_kernel(const float* __restrict__ emb, const float* __restrict__ posq, const int* __restrict__ tok, int ldt, int L, int E,
                          float scale, float* __restrict__ out) {
    const int row = blockIdx.x, b = row / L, j = row % L;
    const float* e = emb + (size_t)tok[b * ldt + j] * E;
    for (int c = threadIdx.x; c < E; c += 256)
        out[(size_t)row * E + c] = scale * e[c] + (j ? posq[(size_t)(j - 1) * E + c] : 0.f);
}

// d emb[v] += scale * sum over the rows whose token is v, rows visited in ascending order (deterministic; one workgroup per token id).
// The B * L token ids are scanned 256 at a time (one compare per thread, the four waves' ballots through LDS) instead of one after the
// other by the whole workgroup — the first form spent 2.4 ms per step on 9 984 dependent loads; the summation order is unchanged.
// Round 3: the rows are cut into gridDim.y chunks of rows_per (a multiple of 256) rows, workgroup (v, c) writes the unscaled sum of ITS rows
// to partial[v][c][E] and embed_bwd_fold_kernel adds the chunks up in ascending order — <pad> is ~45 % of a batch, and ONE workgroup
// walking its 4 500 rows was 0.58 ms of the step.  partial == nullptr (gridDim.y == 1): the single-stage form, straight into demb.
static __global__ __launch_bounds__(256)
void embed_bwd_kernel(const float* __restrict__ dcontent, const int* __restrict__ tok, int ldt, int B, int L, int E, float scale,
                      float* __restrict__ demb, float* __restrict__ partial, int rows_per) {
    __shared__ unsigned long long hits[4];
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(B * L, (int)(blockIdx.y + 1) * rows_per);
    float acc[3] = {0.f, 0.f, 0.f};                              // E <= 768
    for (int base = blockIdx.y * rows_per; base < n; base += 256) {
        const int i = base + tid;
        bool hit = false;
        if (i < n) { const int b = i / L, j = i - b * L; hit = tok[b * ldt + j] == v; }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) hits[wave] = m;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned long long mm = hits[w];                      // the same in every thread
            // eight hits at a time: their row loads go out back to back (a popular id — <pad> is ~45 % of a batch of short labels — used
            // to pay one full memory round trip per row: 2 ms per step), the additions stay in ascending row order
            while (mm) {
                int bit[8];
                int nb = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    bit[q] = mm ? __ffsll((long long)mm) - 1 : -1;
                    if (mm) { mm &= mm - 1; ++nb; }
                }
                float rv[8][3];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float* row = dcontent + (size_t)(base + w * 64 + (bit[q] < 0 ? bit[0] : bit[q])) * E;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int c = tid + 256 * k;
                        rv[q][k] = c < E ? row[c] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < nb) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc[k] += rv[q][k];
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c = tid + 256 * k;
        if (c >= E) continue;
        if (partial) partial[((size_t)v * gridDim.y + blockIdx.y) * E + c] = acc[k];
        else demb[(size_t)v * E + c] += scale * acc[k];
    }
}
// d emb[v] += scale * (the chunks of embed_bwd_kernel in ascending order); one workgroup per token id
static __global__ __launch_bounds__(256)
void embed_bwd_fold_kernel(const float* __restrict__ partial, int chunks, int E, float scale, float* __restrict__ demb) {
    const int v = blockIdx.x;
    for (int c = threadIdx.x; c < E; c += 256) {
        float t = 0.f;
        for (int q = 0; q < chunks; ++q) t += partial[((size_t)v * chunks + q) * E + c];
        demb[(size_t)v * E + c] += scale * t;
    }
}

// d(total loss) / d logits, in place: kept rows (softmax - onehot) * inv_total, ignored rows 0.  One wave per row.
static __global__ __launch_bounds__(256)
void ce_bwd_kernel(float* __restrict__ logits, const int* __restrict__ targets, int rows, int C, int ignore_index, float inv_total) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* row = logits + (size_t)r * C;
    const int tgt = targets[r];
    if (tgt == ignore_index) {
        for (int c = lane; c < C; c += 64) row[c] = 0.f;
        return;
    }
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(row[c] - mx);
    sum = wave_sum(sum);
    const float k = inv_total / sum;
    for (int c = lane; c < C; c += 64) row[c] = expf(row[c] - mx) * k - (c == tgt ? inv_total : 0.f);
}

// loss = sum_k n_k * loss_k / sum_k n_k   (system.py:189-196)
static __global__ void loss_combine_kernel(const float* __restrict__ losses, const int* __restrict__ counts, int K, float* __restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    float num = 0.f; int den = 0;
    for (int k = 0; k < K; ++k) { num += losses[k] * (float)counts[k]; den += counts[k]; }
    *out = num / (float)den;
}

// -------------------------------------------------------------------------------------------------------------------
// Dropout (configs/model/parseq.yaml:21, p = 0.1: model.py:99-102 on the embeddings and the queries, modules.py:33-43,70-79
// inside both attentions, after both attention projections, inside and after the MLP).  A mask is never stored: element `idx`
// of site `site` is kept iff hash(seed, site, idx) >= p * 2^32, a counter-based generator (two rounds of a 32-bit integer
// mixer) that the backward kernels re-evaluate and that oracle/decoder_backward.py restates bit for bit.  The stream differs
// from torch's Philox, so training parity with the reference under dropout is statistical; given the same masks it is exact.
// -------------------------------------------------------------------------------------------------------------------
struct DropSpec {
    unsigned seed_lo, seed_hi;
    unsigned thresh;             // keep iff hash >= thresh; 0 = dropout off
    float scale;                 // 1 / (1 - p)
};

__host__ __device__ __forceinline__ unsigned drop_mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// multiplier of element idx of site: 0 (dropped) or 1 / (1 - p)
__host__ __device__ __forceinline__ float drop_factor(const DropSpec& d, unsigned site, unsigned long long idx) {
    if (d.thresh == 0u) return 1.0f;
    unsigned h = drop_mix((unsigned)idx ^ d.seed_lo);
    h = drop_mix(h + (unsigned)(idx >> 32) * 0x9e3779b9u + site * 0x85ebca6bu + d.seed_hi);
    return h >= d.thresh ? d.scale : 0.0f;
}

// y[i] = (R ? R[i] : 0) + drop(x[i]) (x == y allowed) and y[m][e] = drop(table[m % L][e]) (element index m * E + e: the decoder queries
// pos_queries[:, :L] expanded over the batch), over `passes` permutation passes laid out one after the other ([passes][n_pass] elements):
// pass p draws site `site + 8 p` on the element index WITHIN the pass — what a launch of its own per pass would draw — so that a step may
// run its K passes as one batch of K * B images without changing a single mask bit.  x_shared: x holds one pass ([n_pass]) that every pass reads.
// grid: (ceil(n_pass / 256), passes)
static __global__ __launch_bounds__(256)
void dropout_passes_kernel(const float* x, int x_shared, const float* R, float* y, size_t n_pass, DropSpec d, unsigned site) {
    const size_t li = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (li >= n_pass) return;
    const size_t i = (size_t)blockIdx.y * n_pass + li;
    const float v = x[x_shared ? li : i] * drop_factor(d, site + 8u * blockIdx.y, li);
    y[i] = R ? R[i] + v : v;
}
static __global__ __launch_bounds__(256)
void dropout_rows_passes_kernel(const float* __restrict__ table, int L, int E, float* __restrict__ y, size_t n_pass, DropSpec d, unsigned site) {
    const size_t li = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (li >= n_pass) return;
    const size_t m = li / E, e = li % E;
    y[(size_t)blockIdx.y * n_pass + li] = table[(m % L) * E + e] * drop_factor(d, site + 8u * blockIdx.y, li);
}
// dpre = drop(dact) * gelu'(pre) over [passes][n_pass] elements (grid (ceil(n_pass / 1024), passes), n_pass % 4 == 0): the dropout inside the MLP
// (site `site + 8 p`, element index within the pass — the mask dropout_passes_kernel drew in the forward) and the GELU backward in one pass
static __global__ __launch_bounds__(256)
void gelu_bwd_drop_passes_kernel(const float* __restrict__ pre, const float* dact, float* dpre, size_t n_pass, DropSpec d, unsigned site) {
    const size_t li = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (li >= n_pass) return;
    const size_t i = (size_t)blockIdx.y * n_pass + li;
    const unsigned st = site + 8u * blockIdx.y;
    const float4 v = *reinterpret_cast<const float4*>(pre + i), g = *reinterpret_cast<const float4*>(dact + i);
    *reinterpret_cast<float4*>(dpre + i) = make_float4(g.x * drop_factor(d, st, li) * gelu_grad(v.x), g.y * drop_factor(d, st, li + 1) * gelu_grad(v.y),
                                                       g.z * drop_factor(d, st, li + 2) * gelu_grad(v.z), g.w * drop_factor(d, st, li + 3) * gelu_grad(v.w));
}
// y[li] (+)= sum over the passes, in ascending order, of drop(x[p][li]) (d.thresh == 0: a plain sum of the passes)
static __global__ __launch_bounds__(256)
void dropout_sum_passes_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n_pass, int passes, DropSpec d, unsigned site, int accumulate) {
    const size_t li = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (li >= n_pass) return;
    float t = accumulate ? y[li] : 0.f;
    for (int p = 0; p < passes; ++p) t += x[(size_t)p * n_pass + li] * drop_factor(d, site + 8u * (unsigned)p, li);
    y[li] = t;
}
// y[i] (+)= sum over the passes of x[p][i], four elements per thread (n_pass a multiple of 4, 16-byte aligned)
static __global__ __launch_bounds__(256)
void sum_passes_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n_pass, int passes, int accumulate) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n_pass) return;
    float4 t = accumulate ? *reinterpret_cast<const float4*>(y + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < passes; ++p) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)p * n_pass + i);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(y + i) = t;
}

// -------------------------------------------------------------------------------------------------------------------
// Soft-max attention (decoder: head width 32, encoder: 64), one workgroup per (image b, head h), operands in LDS, the
// queries walked in blocks of 32 rows (the key / value gradients of a head stay in registers across the blocks).
//   q row (b, l): q + b * q_bstride + l * ldq + HD h        (q_bstride = 0: the queries are shared by the batch)
//   k / v row (b, j): k|v + (b * Lk + j) * ldkv + HD h
//   key j of query l is masked when qmask[l * Lk + j] or kmask[b * ldkm + j] (either pointer may be null)
// Every query must keep at least one key (true on this path: <bos> is never masked).  Lk * HD <= 8192.
// -------------------------------------------------------------------------------------------------------------------
struct TrainAttnArgs {
    const float* q; long q_bstride; int ldq;
    const float* k; const float* v; int ldkv;
    const unsigned char* qmask; const unsigned char* kmask; int ldkm;
    float* o; int ldo;                                           // forward output, row (b, l): o + (b * Lq + l) * ldo + HD h
    bf16_t* o16 = nullptr;                                       // train_attn_bf16_kernel forward: write o here as bf16 INSTEAD (same layout)
    bf16_t* dq16 = nullptr; bf16_t* dk16 = nullptr; bf16_t* dv16 = nullptr;      // train_attn_bf16_kernel backward: write dq / dk / dv here as bf16
                                                                 // INSTEAD (layouts of dq / dk / dv; no kv_accumulate)
    const float* d_o;                                            // backward: gradient of o (same layout as o)
    float* dq; int lddq;                                         // backward: stored, row (b, l) even when q is shared
    float* dk; float* dv; int lddkv;                             // backward: layout of k / v; accumulated into if kv_accumulate
    int kv_accumulate;
    int Lq, Lk, H;
    float scale;
    DropSpec drop; unsigned drop_site;                           // dropout on the probabilities, element ((b H + h) Lq + l) Lk + j
    // Several permutation passes in one launch (train_attn_kernel and train_attn_dec_bf16_kernel; 0 = off): the batch is [passes][pass_B]
    // images, image index b = p * pass_B + bl.  q / o / d_o / dq / dk / dv rows are indexed by b; the key-padding mask and the dropout
    // element index by bl (what pass p's own launch would use), the query mask is qmask + p * qmask_pstride, the dropout site
    // drop_site + p * site_pstride; kv_shared: k / v rows are image bl's for every pass (cross-attention over the encoder memory).
    int pass_B = 0; long qmask_pstride = 0; unsigned site_pstride = 0; int kv_shared = 0;
    // train_attn_dec_bf16_kernel only, with pass_B and kv_shared: the launch has pass_B * H workgroups and each walks pass_loop passes of its
    // image (K / V staged once, dK / dV summed over the passes in the accumulators and stored once, rows of image bl)
    int pass_loop = 0;
};
// (image over all passes, head) of a workgroup -> what the pass-batched launch indexes with; pass_B == 0 reduces to the plain launch
struct TrainAttnIdx { int bf, bl, bkv, h; const unsigned char* qmask; unsigned site; unsigned long long dblock; };
__device__ __forceinline__ TrainAttnIdx train_attn_idx(const TrainAttnArgs& a) {
    TrainAttnIdx x;
    x.bf = blockIdx.x / a.H; x.h = blockIdx.x % a.H;
    const int p = a.pass_B ? x.bf / a.pass_B : 0;
    x.bl = a.pass_B ? x.bf - p * a.pass_B : x.bf;
    x.bkv = a.kv_shared ? x.bl : x.bf;
    x.qmask = a.qmask ? a.qmask + (size_t)p * a.qmask_pstride : nullptr;
    x.site = a.drop_site + (unsigned)p * a.site_pstride;
    x.dblock = (unsigned long long)x.bl * a.H + x.h;
    return x;
}

constexpr int TA_QB = 32, TA_NACC = 32;

__host__ __device__ inline size_t train_attn_lds_floats(int Lq, int Lk, int hd, bool backward) {
    const int nq = Lq < TA_QB ? Lq : TA_QB;
    return (size_t)2 * Lk * (hd + 1) + (size_t)(backward ? 2 : 1) * nq * (hd + 1) + (size_t)(backward ? 3 : 1) * nq * (Lk + 1);
}

template <bool BACKWARD, int HD>
__global__ __launch_bounds__(256)
void train_attn_kernel(const TrainAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ta_smem[];
    constexpr int PAD = HD + 1;
    const int Lq = a.Lq, Lk = a.Lk, ldp = Lk + 1;
    const int nqmax = Lq < TA_QB ? Lq : TA_QB;
    float* Ks = ta_smem;                          // [Lk][PAD]
    float* Vs = Ks + (size_t)Lk * PAD;            // [Lk][PAD]
    float* Qs = Vs + (size_t)Lk * PAD;            // [nq][PAD]
    float* P = Qs + (size_t)nqmax * PAD;          // [nq][Lk + 1]
    float* dOs = P + (size_t)nqmax * ldp;         // [nq][PAD]      (backward only)
    float* dS = dOs + (size_t)nqmax * PAD;        // [nq][Lk + 1]   (backward only)
    float* PD = dS + (size_t)nqmax * ldp;         // [nq][Lk + 1]   (backward only) probabilities after dropout
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TrainAttnIdx ix = train_attn_idx(a);
    const int b = ix.bf, h = ix.h;      // b: the image over all passes (rows of q / o / d_o / dq / dk / dv)

    for (int idx = tid; idx < Lk * HD; idx += 256) {
        const int j = idx / HD, d = idx % HD;
        const size_t g = ((size_t)ix.bkv * Lk + j) * a.ldkv + h * HD + d;
        Ks[j * PAD + d] = a.k[g];
        Vs[j * PAD + d] = a.v[g];
    }
    float gk[TA_NACC], gv[TA_NACC];
#pragma unroll
    for (int i = 0; i < TA_NACC; ++i) { gk[i] = 0.f; gv[i] = 0.f; }

    for (int q0 = 0; q0 < Lq; q0 += TA_QB) {
        const int nq = (Lq - q0) < TA_QB ? (Lq - q0) : TA_QB;
        __syncthreads();                          // the previous block's readers are done with Qs / P / dOs / dS
        for (int idx = tid; idx < nq * HD; idx += 256) {
            const int l = idx / HD, d = idx % HD;
            Qs[l * PAD + d] = a.q[(size_t)b * a.q_bstride + (size_t)(q0 + l) * a.ldq + h * HD + d];
            if (BACKWARD) dOs[l * PAD + d] = a.d_o[((size_t)b * Lq + q0 + l) * a.ldo + h * HD + d];
        }
        __syncthreads();
        // scores (and, backward, dP = dO V^T)
        for (int idx = tid; idx < nq * Lk; idx += 256) {
            const int l = idx / Lk, j = idx % Lk;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                s = fmaf(Qs[l * PAD + d], Ks[j * PAD + d], s);
                if (BACKWARD) dp = fmaf(dOs[l * PAD + d], Vs[j * PAD + d], dp);
            }
            const bool masked = (ix.qmask && ix.qmask[(size_t)(q0 + l) * Lk + j]) || (a.kmask && a.kmask[(size_t)ix.bl * a.ldkm + j]);
            P[l * ldp + j] = masked ? -INFINITY : s * a.scale;
            if (BACKWARD) dS[l * ldp + j] = dp;
        }
        __syncthreads();
        // soft-max over the keys, one wave per query row (and, backward, dS = P * (dP - sum_j dP P) * scale)
        for (int l = wave; l < nq; l += 4) {
            float mx = -INFINITY;
            for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, P[l * ldp + j]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = lane; j < Lk; j += 64) { const float e = expf(P[l * ldp + j] - mx); P[l * ldp + j] = e; sum += e; }
            const float inv = 1.0f / wave_sum(sum);
            const unsigned long long row = (ix.dblock * Lq + q0 + l) * Lk;
            float dot = 0.f;
            for (int j = lane; j < Lk; j += 64) {
                const float p = P[l * ldp + j] * inv;
                const float f = drop_factor(a.drop, ix.site, row + j);
                if (BACKWARD) {
                    P[l * ldp + j] = p;                       // soft-max output
                    PD[l * ldp + j] = p * f;                  // what multiplied V
                    const float dp = dS[l * ldp + j] * f;     // gradient w.r.t. the soft-max output
                    dS[l * ldp + j] = dp;
                    dot += dp * p;
                } else {
                    P[l * ldp + j] = p * f;
                }
            }
            if (BACKWARD) {
                dot = wave_sum(dot);
                for (int j = lane; j < Lk; j += 64) dS[l * ldp + j] = P[l * ldp + j] * (dS[l * ldp + j] - dot) * a.scale;
            }
        }
        __syncthreads();
        if (!BACKWARD) {
            for (int idx = tid; idx < nq * HD; idx += 256) {
                const int l = idx / HD, d = idx % HD;
                float o = 0.f;
                for (int j = 0; j < Lk; ++j) o = fmaf(P[l * ldp + j], Vs[j * PAD + d], o);
                a.o[((size_t)b * Lq + q0 + l) * a.ldo + h * HD + d] = o;
            }
        } else {
            for (int idx = tid; idx < nq * HD; idx += 256) {
                const int l = idx / HD, d = idx % HD;
                float g = 0.f;
                for (int j = 0; j < Lk; ++j) g = fmaf(dS[l * ldp + j], Ks[j * PAD + d], g);
                a.dq[((size_t)b * Lq + q0 + l) * a.lddq + h * HD + d] = g;
            }
#pragma unroll
            for (int i = 0; i < TA_NACC; ++i) {
                const int idx = tid + 256 * i;
                if (idx < Lk * HD) {
                    const int j = idx / HD, d = idx % HD;
                    for (int l = 0; l < nq; ++l) {
                        gk[i] = fmaf(dS[l * ldp + j], Qs[l * PAD + d], gk[i]);
                        gv[i] = fmaf(PD[l * ldp + j], dOs[l * PAD + d], gv[i]);
                    }
                }
            }
        }
    }
    if (BACKWARD) {
#pragma unroll
        for (int i = 0; i < TA_NACC; ++i) {
            const int idx = tid + 256 * i;
            if (idx < Lk * HD) {
                const int j = idx / HD, d = idx % HD;
                const size_t g = ((size_t)b * Lk + j) * a.lddkv + h * HD + d;
                a.dk[g] = a.kv_accumulate ? a.dk[g] + gk[i] : gk[i];
                a.dv[g] = a.kv_accumulate ? a.dv[g] + gv[i] : gv[i];
            }
        }
    }
}

// The encoder's attention (head width 64, no masks, no dropout, Lq % 32 == 0, Lk % 16 == 0, Lk <= 128) with its five products on
// the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): same LDS residency, query blocks and register-held dK / dV as above.
// MFMA conventions as in mfma_sgemm_kernel: lane (r16 = lane & 15, g = lane >> 4) feeds A[row r16][k g] and B[k g][col r16] and
// receives D[row 4 g + r][col r16], r = 0..3.
template <bool BACKWARD>
__global__ __launch_bounds__(256)
void train_attn_mfma_kernel(const TrainAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ta_smem[];
    constexpr int HD = 64, PAD = HD + 1, QB = 32, MAXJT = 8;
    const int Lq = a.Lq, Lk = a.Lk, ldp = Lk + 1, njt = Lk / 16;
    float* Ks = ta_smem;                          // [Lk][PAD]
    float* Vs = Ks + (size_t)Lk * PAD;            // [Lk][PAD]
    float* Qs = Vs + (size_t)Lk * PAD;            // [QB][PAD]
    float* P = Qs + (size_t)QB * PAD;             // [QB][Lk + 1]
    float* dOs = P + (size_t)QB * ldp;            // [QB][PAD]      (backward only)
    float* dS = dOs + (size_t)QB * PAD;           // [QB][Lk + 1]   (backward only)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;

    for (int idx = tid; idx < Lk * HD; idx += 256) {
        const int j = idx / HD, d = idx % HD;
        const size_t gi = ((size_t)b * Lk + j) * a.ldkv + h * HD + d;
        Ks[j * PAD + d] = a.k[gi];
        Vs[j * PAD + d] = a.v[gi];
    }
    f32x4 gk[MAXJT], gv[MAXJT];                   // dK / dV tiles (key tile jt, head columns [16 wave, +16)) of this wave
#pragma unroll
    for (int i = 0; i < MAXJT; ++i) { gk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int q0 = 0; q0 < Lq; q0 += QB) {
        __syncthreads();
        for (int idx = tid; idx < QB * HD; idx += 256) {
            const int l = idx / HD, d = idx % HD;
            Qs[l * PAD + d] = a.q[(size_t)b * a.q_bstride + (size_t)(q0 + l) * a.ldq + h * HD + d];
            if (BACKWARD) dOs[l * PAD + d] = a.d_o[((size_t)b * Lq + q0 + l) * a.ldo + h * HD + d];
        }
        __syncthreads();
        // S = Q K^T (and dP = dO V^T): key tiles wave, wave + 4 for both 16-row halves of the query block
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int jt = wave + 4 * jj;
            if (jt < njt) {
                f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                f32x4 pacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4
                for (int ks = 0; ks < HD / 4; ++ks) {
                    const float kb = Ks[(16 * jt + r16) * PAD + 4 * ks + g];
                    const float q0v = Qs[r16 * PAD + 4 * ks + g], q1v = Qs[(16 + r16) * PAD + 4 * ks + g];
                    sacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(q0v, kb, sacc[0], 0, 0, 0);
                    sacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(q1v, kb, sacc[1], 0, 0, 0);
                    if (BACKWARD) {
                        const float vb = Vs[(16 * jt + r16) * PAD + 4 * ks + g];
                        const float o0 = dOs[r16 * PAD + 4 * ks + g], o1 = dOs[(16 + r16) * PAD + 4 * ks + g];
                        pacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o0, vb, pacc[0], 0, 0, 0);
                        pacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o1, vb, pacc[1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int lt = 0; lt < 2; ++lt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int at = (16 * lt + 4 * g + r) * ldp + 16 * jt + r16;
                        P[at] = sacc[lt][r] * a.scale;
                        if (BACKWARD) dS[at] = pacc[lt][r];
                    }
            }
        }
        __syncthreads();
        // soft-max over the keys, one wave per query row (and, backward, dS = P * (dP - sum_j dP P) * scale)
        for (int l = wave; l < QB; l += 4) {
            float mx = -INFINITY;
            for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, P[l * ldp + j]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = lane; j < Lk; j += 64) { const float e = expf(P[l * ldp + j] - mx); P[l * ldp + j] = e; sum += e; }
            const float inv = 1.0f / wave_sum(sum);
            float dot = 0.f;
            for (int j = lane; j < Lk; j += 64) {
                const float p = P[l * ldp + j] * inv;
                P[l * ldp + j] = p;
                if (BACKWARD) dot += dS[l * ldp + j] * p;
            }
            if (BACKWARD) {
                dot = wave_sum(dot);
                for (int j = lane; j < Lk; j += 64) dS[l * ldp + j] = P[l * ldp + j] * (dS[l * ldp + j] - dot) * a.scale;
            }
        }
        __syncthreads();
        // O = P V (forward) / dQ = dS K (backward): head columns [16 wave, +16) for both halves of the query block
        {
            const float* L_ = BACKWARD ? dS : P;
            const float* R_ = BACKWARD ? Ks : Vs;
            f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4
            for (int ks = 0; ks < Lk / 4; ++ks) {
                const float rb = R_[(4 * ks + g) * PAD + 16 * wave + r16];
                const float l0 = L_[r16 * ldp + 4 * ks + g], l1 = L_[(16 + r16) * ldp + 4 * ks + g];
                oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(l0, rb, oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(l1, rb, oacc[1], 0, 0, 0);
            }
            float* dst = BACKWARD ? a.dq : a.o;
            const int ldd = BACKWARD ? a.lddq : a.ldo;
#pragma unroll
            for (int lt = 0; lt < 2; ++lt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[((size_t)b * Lq + q0 + 16 * lt + 4 * g + r) * ldd + h * HD + 16 * wave + r16] = oacc[lt][r];
        }
        if (BACKWARD) {
            // dK += dS^T Q, dV += P^T dO over the 32 query rows of the block: every key tile, head columns [16 wave, +16)
            // (ks is NOT unrolled: with all 64 (ks, jt) bodies in flight the scheduler hoists 128 LDS loads and spills 2000 VGPRs)
#pragma unroll 1
            for (int ks = 0; ks < QB / 4; ++ks) {
                const float qb = Qs[(4 * ks + g) * PAD + 16 * wave + r16];
                const float ob = dOs[(4 * ks + g) * PAD + 16 * wave + r16];
#pragma unroll
                for (int jt = 0; jt < MAXJT; ++jt) {
                    if (jt < njt) {
                        const float sa = dS[(4 * ks + g) * ldp + 16 * jt + r16];
                        const float pa = P[(4 * ks + g) * ldp + 16 * jt + r16];
                        gk[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa, qb, gk[jt], 0, 0, 0);
                        gv[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, ob, gv[jt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (BACKWARD) {
#pragma unroll
        for (int jt = 0; jt < MAXJT; ++jt) {
            if (jt < njt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t gi = ((size_t)b * Lk + 16 * jt + 4 * g + r) * a.lddkv + h * HD + 16 * wave + r16;
                    a.dk[gi] = a.kv_accumulate ? a.dk[gi] + gk[jt][r] : gk[jt][r];
                    a.dv[gi] = a.kv_accumulate ? a.dv[gi] + gv[jt][r] : gv[jt][r];
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------
// Encoder attention of the training step in the bf16-operand mode (train_precision = bf16): 128 tokens, head width 64, no masks, no
// dropout — the five products S = Q K^T, O = P V, dP = dO V^T, dQ = dS K, dK = dS^T Q, dV = P^T dO on v_mfma_f32_16x16x32_bf16 with
// fp32 accumulation, soft-max and its backward in fp32 registers (the fp32 mode keeps train_attn_mfma_kernel: exact 16x16x4 products).
// One workgroup of four waves per (image, head); the queries are walked in two blocks of 64, wave w owns the block's query tile w.
//   S^T = K Q^T is computed (A = K rows, B = Q rows): lane (r16, g) then holds, for ITS query l = r16, the scores of keys
//   16 jt + 4 g + r — the soft-max over the keys is a reduction over the lane's registers and the four lane groups (rows4_max / sum),
//   and the probabilities are already the B operand (k-slots = keys) of O^T = V^T P^T and, backward, dS^T of dQ^T = K^T dS^T: no LDS
//   round trip between the score product and the products that contract over the keys (the inference kernel's trick,
//   encoder_attn.h).  The products that contract over the QUERIES (dK, dV) need P^T / dS^T with lanes indexed by key: those go
//   through LDS as bf16 [key][query] images, written two bytes at a time; the operands indexed by head column with a token k-axis
//   (V^T, K^T, Q^T, dO^T) are staged transposed once.
// Rounding points of the mode: q, k, v, dO (operands), p and dS (operands of the second-level products) are rounded to bf16;
// everything else is fp32.  MFMA conventions as everywhere (common.h mma16): result lane (r16, g) holds D[row 4 g + r][col r16].
// -------------------------------------------------------------------------------------------------------------------
constexpr int TB_N = 128, TB_HD = 64, TB_QB = 64;
constexpr int TB_RP = TB_HD + 8;          // pitch (elements) of [token][d] images: 144 B rows, conflict-free ds_read_b128
constexpr int TB_TP = TB_N + 8;           // pitch of [d][key] images (V^T, K^T)
constexpr int TB_PP = TB_QB + 8;          // pitch of [key][query in block] / [d][query in block] images
constexpr size_t train_attn_bf16_lds(bool backward) {
    return sizeof(bf16_t) * (backward ? (size_t)2 * TB_N * TB_RP + (size_t)TB_HD * TB_TP + (size_t)2 * TB_QB * TB_RP + (size_t)2 * TB_HD * TB_PP + (size_t)2 * TB_N * TB_PP
                                      : (size_t)TB_N * TB_RP + (size_t)TB_HD * TB_TP + (size_t)TB_QB * TB_RP);
}

// rows [r0, r0 + NR) of a [*, 64] fp32 matrix (row stride ld) -> bf16 row-major image (pitch TB_RP) and / or transposed image
// (dst_t[d][row - r0], pitch tp); all 256 threads, 16-byte global loads
// All of a call's loads are issued before the first conversion (NR is a compile-time row count: the loop form was compiled to one load,
// `s_waitcnt vmcnt(0)`, its LDS stores, next load ... — a full memory round trip per 16 bytes and thread, most of the first version's time).
template <bool ROWS, bool TRANS, int NR>
__device__ __forceinline__ void tb_stage(const float* __restrict__ src, long ld, bf16_t* dst_r, bf16_t* dst_t, int tp, int tid) {
    constexpr int IT = NR * (TB_HD / 4) / 256;
    float4 ld4[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + 256 * it;
        ld4[it] = *reinterpret_cast<const float4*>(src + (size_t)(idx >> 4) * ld + (idx & 15) * 4);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + 256 * it, row = idx >> 4, c4 = (idx & 15) * 4;
        const float4 v = ld4[it];
        const bf16_t e0 = static_cast<bf16_t>(v.x), e1 = static_cast<bf16_t>(v.y), e2 = static_cast<bf16_t>(v.z), e3 = static_cast<bf16_t>(v.w);
        if constexpr (ROWS) {
            bf16x4 o; o[0] = e0; o[1] = e1; o[2] = e2; o[3] = e3;
            *reinterpret_cast<bf16x4*>(dst_r + row * TB_RP + c4) = o;
        }
        if constexpr (TRANS) {
            dst_t[(c4 + 0) * tp + row] = e0; dst_t[(c4 + 1) * tp + row] = e1;
            dst_t[(c4 + 2) * tp + row] = e2; dst_t[(c4 + 3) * tp + row] = e3;
        }
    }
}

// A fragment of a [d][key] image for the key k-slot order of the S^T accumulators: lane (d = row0 + r16, g), k-step kk (32 keys):
// slots 0-3 = keys 32 kk + 4 g + [0, 4), slots 4-7 = keys 32 kk + 16 + 4 g + [0, 4)
__device__ __forceinline__ Frag<bf16_t> tb_keyslot_frag(const bf16_t* img, int row, int kk, int g) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(img + row * TB_TP + 32 * kk + 4 * g);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(img + row * TB_TP + 32 * kk + 16 + 4 * g);
    Frag<bf16_t> f;
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3]; f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
    return f;
}

template <bool BACKWARD>
__global__ __launch_bounds__(256)
void train_attn_bf16_kernel(const TrainAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tb_smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(tb_smem);               // [128][TB_RP]  K
    bf16_t* Vs = Ks + TB_N * TB_RP;                                 // backward: [128][TB_RP] V
    bf16_t* XT = BACKWARD ? Vs + TB_N * TB_RP : Ks + TB_N * TB_RP;  // [64][TB_TP]  forward: V^T; backward: K^T
    bf16_t* Qs = XT + TB_HD * TB_TP;                                // [64][TB_RP]  Q block
    bf16_t* dOs = Qs + TB_QB * TB_RP;                               // backward: [64][TB_RP] dO block
    bf16_t* Qt = dOs + TB_QB * TB_RP;                               // backward: [64 d][TB_PP] Q^T block
    bf16_t* dOt = Qt + TB_HD * TB_PP;                               // backward: [64 d][TB_PP] dO^T block
    bf16_t* Pt = dOt + TB_HD * TB_PP;                               // backward: [128 key][TB_PP] P^T
    bf16_t* dSt = Pt + TB_N * TB_PP;                                // backward: [128 key][TB_PP] dS^T
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const float sl2 = a.scale * 1.44269504088896340736f;           // scale * log2(e): p = exp2((s - max) * sl2)

    const float* kg = a.k + (size_t)b * TB_N * a.ldkv + h * TB_HD;
    const float* vg = a.v + (size_t)b * TB_N * a.ldkv + h * TB_HD;
    if constexpr (BACKWARD) {
        tb_stage<true, true, TB_N>(kg, a.ldkv, Ks, XT, TB_TP, tid);
        tb_stage<true, false, TB_N>(vg, a.ldkv, Vs, nullptr, 0, tid);
    } else {
        tb_stage<true, false, TB_N>(kg, a.ldkv, Ks, nullptr, 0, tid);
        tb_stage<false, true, TB_N>(vg, a.ldkv, nullptr, XT, TB_TP, tid);
    }
    f32x4 gk[2][4], gv[2][4];                    // backward: dK / dV of key tiles 2 wave + {0, 1}, head-column tiles 0..3
    if constexpr (BACKWARD) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { gk[jj][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; gv[jj][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }

    for (int q0 = 0; q0 < TB_N; q0 += TB_QB) {
        __syncthreads();                         // the previous block's readers of Qs / dOs / Qt / dOt / Pt / dSt are done
        const float* qg = a.q + (size_t)b * a.q_bstride + (size_t)q0 * a.ldq + h * TB_HD;
        if constexpr (BACKWARD) {
            tb_stage<true, true, TB_QB>(qg, a.ldq, Qs, Qt, TB_PP, tid);
            tb_stage<true, true, TB_QB>(a.d_o + ((size_t)b * TB_N + q0) * a.ldo + h * TB_HD, a.ldo, dOs, dOt, TB_PP, tid);
        } else {
            tb_stage<true, false, TB_QB>(qg, a.ldq, Qs, nullptr, 0, tid);
        }
        __syncthreads();
        // ---- S^T (and dP^T) of this wave's 16 queries against all 128 keys
        Frag<bf16_t> qf[2], of[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks].v = *reinterpret_cast<const bf16x8*>(Qs + (16 * wave + r16) * TB_RP + 32 * ks + 8 * g);
            if constexpr (BACKWARD) of[ks].v = *reinterpret_cast<const bf16x8*>(dOs + (16 * wave + r16) * TB_RP + 32 * ks + 8 * g);
        }
        f32x4 sacc[8], pacc[8];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            sacc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BACKWARD) pacc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<bf16_t> kf;
                kf.v = *reinterpret_cast<const bf16x8*>(Ks + (16 * jt + r16) * TB_RP + 32 * ks + 8 * g);
                mma16(sacc[jt], kf, qf[ks]);
                if constexpr (BACKWARD) {
                    Frag<bf16_t> vf;
                    vf.v = *reinterpret_cast<const bf16x8*>(Vs + (16 * jt + r16) * TB_RP + 32 * ks + 8 * g);
                    mma16(pacc[jt], vf, of[ks]);
                }
            }
        }
        // ---- soft-max over the keys of query l = r16: registers, then the four lane groups
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[jt][r]);
        mx = rows4_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f((sacc[jt][r] - mx) * sl2); sacc[jt][r] = e; sum += e; }
        sum = rows4_sum(sum);
        const float inv = 1.0f / sum;
        if constexpr (!BACKWARD) {
            // ---- O^T = V^T P^T with the un-normalised probabilities as the B operand; 1 / sum applied to the result
            Frag<bf16_t> pf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) pf[kk].v[s8] = static_cast<bf16_t>(sacc[2 * kk + (s8 >> 2)][s8 & 3]);
            float* og = a.o + ((size_t)b * TB_N + q0 + 16 * wave + r16) * a.ldo + h * TB_HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) mma16(oacc, tb_keyslot_frag(XT, 16 * dt + r16, kk, g), pf[kk]);
                // oacc[r] = O[query r16][d = 16 dt + 4 g + r]
                if (a.o16) {
                    union { uint2 u; bf16_t e[4]; } h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h.e[r] = static_cast<bf16_t>(oacc[r] * inv);
                    *reinterpret_cast<uint2*>(a.o16 + (og - a.o) + 16 * dt + 4 * g) = h.u;
                } else
                *reinterpret_cast<f32x4*>(og + 16 * dt + 4 * g) = oacc * inv;
            }
        } else {
            // ---- dS^T = P^T * (dP^T - sum_j P dP) * scale; P^T and dS^T to LDS ([key][query]) for the products over the queries
            float dot = 0.f;
#pragma unroll
            for (int jt = 0; jt < 8; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sacc[jt][r] *= inv; dot += sacc[jt][r] * pacc[jt][r]; }
            dot = rows4_sum(dot);
#pragma unroll
            for (int jt = 0; jt < 8; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = sacc[jt][r];
                    const float ds = p * (pacc[jt][r] - dot) * a.scale;
                    pacc[jt][r] = ds;
                    Pt[(16 * jt + 4 * g + r) * TB_PP + 16 * wave + r16] = static_cast<bf16_t>(p);
                    dSt[(16 * jt + 4 * g + r) * TB_PP + 16 * wave + r16] = static_cast<bf16_t>(ds);
                }
            // ---- dQ^T = K^T dS^T with dS^T straight from the registers (k-slots = keys)
            Frag<bf16_t> df[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) df[kk].v[s8] = static_cast<bf16_t>(pacc[2 * kk + (s8 >> 2)][s8 & 3]);
            float* dqg = a.dq + ((size_t)b * TB_N + q0 + 16 * wave + r16) * a.lddq + h * TB_HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 qacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) mma16(qacc, tb_keyslot_frag(XT, 16 * dt + r16, kk, g), df[kk]);
                if (a.dq16) {
                    union { u32x2 u; bf16_t e[4]; } hq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hq.e[r] = static_cast<bf16_t>(qacc[r]);
                    *reinterpret_cast<u32x2*>(a.dq16 + (dqg - a.dq) + 16 * dt + 4 * g) = hq.u;
                } else
                *reinterpret_cast<f32x4*>(dqg + 16 * dt + 4 * g) = qacc;
            }
            __syncthreads();                     // P^T / dS^T of all 64 queries of the block are in LDS
            // ---- dV += P^T dO, dK += dS^T Q over the block's 64 queries: key tiles 2 wave + jj, every head-column tile
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<bf16_t> pa[2], sa[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    pa[jj].v = *reinterpret_cast<const bf16x8*>(Pt + (16 * (2 * wave + jj) + r16) * TB_PP + 32 * ks + 8 * g);
                    sa[jj].v = *reinterpret_cast<const bf16x8*>(dSt + (16 * (2 * wave + jj) + r16) * TB_PP + 32 * ks + 8 * g);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag<bf16_t> ob, qb;
                    ob.v = *reinterpret_cast<const bf16x8*>(dOt + (16 * dt + r16) * TB_PP + 32 * ks + 8 * g);
                    qb.v = *reinterpret_cast<const bf16x8*>(Qt + (16 * dt + r16) * TB_PP + 32 * ks + 8 * g);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        mma16(gv[jj][dt], ob, pa[jj]);       // transposed tiles: gv[r] = dV[key 16 jt + r16][d = 16 dt + 4 g + r] (16-byte stores below)
                        mma16(gk[jj][dt], qb, sa[jj]);
                    }
                }
            }
        }
    }
    if constexpr (BACKWARD) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const size_t gi = ((size_t)b * TB_N + 16 * (2 * wave + jj) + r16) * a.lddkv + h * TB_HD + 16 * dt + 4 * g;
                if (a.dk16) {
                    union { u32x2 u; bf16_t e[4]; } hk, hv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { hk.e[r] = static_cast<bf16_t>(gk[jj][dt][r]); hv.e[r] = static_cast<bf16_t>(gv[jj][dt][r]); }
                    *reinterpret_cast<u32x2*>(a.dk16 + gi) = hk.u;
                    *reinterpret_cast<u32x2*>(a.dv16 + gi) = hv.u;
                    continue;
                }
                f32x4 ok_ = f32x4{0.f, 0.f, 0.f, 0.f}, ov_ = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.kv_accumulate) { ok_ = *reinterpret_cast<const f32x4*>(a.dk + gi); ov_ = *reinterpret_cast<const f32x4*>(a.dv + gi); }
                *reinterpret_cast<f32x4*>(a.dk + gi) = ok_ + gk[jj][dt];
                *reinterpret_cast<f32x4*>(a.dv + gi) = ov_ + gv[jj][dt];
            }
    }
}

// -------------------------------------------------------------------------------------------------------------------
// Decoder attention of the training step in the bf16-operand mode: head width 32, <= 32 queries (one block), <= 128 keys, torch-style
// boolean masks (query x key and per-image key padding) and dropout on the probabilities — the same arrangement as
// train_attn_bf16_kernel (S^T = K Q^T, lane = query; probabilities as the B operand of the key-contracting products; P^T / dS^T
// through LDS for dK / dV) at the decoder's shapes: one workgroup of TWO waves per (image, head), wave w owns query tile w and the
// key tiles jt = w (mod 2) of dK / dV.  Queries past Lq and keys past Lk are zero rows (keys additionally masked), so they
// contribute nothing and are never stored.  Dropout: the counter-based generator of train_attn_kernel, same element index
// ((b H + h) Lq + l) Lk + j, so the two kernels drop the same probabilities.
// -------------------------------------------------------------------------------------------------------------------
constexpr int TD_HD = 32, TD_Q = 32, TD_K = 128;
constexpr int TD_RP = TD_HD + 8;          // pitch of [token][d] images (80-byte rows: conflict-free ds_read_b128)
constexpr int TD_TP = TD_K + 8;           // pitch of [d][key] images at 128 keys (KT 16-key tiles: 16 KT + 8)
constexpr int TD_PP = TD_Q + 8;           // pitch of [key][query] and [d][query] images
// KT: the 16-key tiles the instantiation holds — 8 (128 keys: the cross-attention over the encoder memory) or 2 (32 keys: the self-attention
// over <= 26 context positions, whose workgroups then take a quarter of the LDS and a third of the registers: 27 648 of them per step
// batch, each a chain of dependent memory round trips, so what the launch needs is more of them resident)
constexpr size_t train_attn_dec_lds(bool backward, int KT = 8) {
    const size_t K_ = 16 * (size_t)KT, TP_ = K_ + 8;
    return sizeof(bf16_t) * (backward ? 2 * K_ * TD_RP + (size_t)TD_HD * TP_ + (size_t)2 * TD_Q * TD_RP + (size_t)2 * TD_HD * TD_PP + 2 * K_ * TD_PP
                                      : K_ * TD_RP + (size_t)TD_HD * TP_ + (size_t)TD_Q * TD_RP);
}

// rows [0, NRP) of the images; rows >= nr are zero.  src row r at src + r * ld (32 floats used).  128 threads.
// MAXR: compile-time bound of NRP; all loads of a call are issued before the first conversion (see tb_stage)
template <bool ROWS, bool TRANS, int MAXR>
__device__ __forceinline__ void td_stage(const float* __restrict__ src, long ld, int nr, int NRP, bf16_t* dst_r, bf16_t* dst_t, int tp, int tid) {
    constexpr int IT = MAXR * (TD_HD / 4) / 128;
    float4 ld4[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + 128 * it, row = idx >> 3;
        ld4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nr) ld4[it] = *reinterpret_cast<const float4*>(src + (size_t)row * ld + (idx & 7) * 4);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = tid + 128 * it, row = idx >> 3, c4 = (idx & 7) * 4;
        if (row >= NRP) continue;
        const float4 v = ld4[it];
        const bf16_t e0 = static_cast<bf16_t>(v.x), e1 = static_cast<bf16_t>(v.y), e2 = static_cast<bf16_t>(v.z), e3 = static_cast<bf16_t>(v.w);
        if constexpr (ROWS) {
            bf16x4 o; o[0] = e0; o[1] = e1; o[2] = e2; o[3] = e3;
            *reinterpret_cast<bf16x4*>(dst_r + row * TD_RP + c4) = o;
        }
        if constexpr (TRANS) {
            dst_t[(c4 + 0) * tp + row] = e0; dst_t[(c4 + 1) * tp + row] = e1;
            dst_t[(c4 + 2) * tp + row] = e2; dst_t[(c4 + 3) * tp + row] = e3;
        }
    }
}
template <int TP>
__device__ __forceinline__ Frag<bf16_t> td_keyslot_frag(const bf16_t* img, int row, int kk, int g) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(img + row * TP + 32 * kk + 4 * g);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(img + row * TP + 32 * kk + 16 + 4 * g);
    Frag<bf16_t> f;
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3]; f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
    return f;
}

template <bool BACKWARD, int KT = 8>
__global__ __launch_bounds__(128)
void train_attn_dec_bf16_kernel(const TrainAttnArgs a) {
    static_assert(KT == 2 || KT == 8, "key tiles per instantiation");
    constexpr int KMAX = 16 * KT, TP = KMAX + 8, KK = KT / 2, JJ = KT / 2;      // keys held, pitch of the [d][key] image, 32-key k-steps, key tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char td_smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(td_smem);               // [KMAX][TD_RP]  K
    bf16_t* Vs = Ks + KMAX * TD_RP;                                 // backward: [KMAX][TD_RP] V
    bf16_t* XT = BACKWARD ? Vs + KMAX * TD_RP : Ks + KMAX * TD_RP;  // [32][TP]  forward: V^T; backward: K^T
    bf16_t* Qs = XT + TD_HD * TP;                                   // [32][TD_RP]  Q
    bf16_t* dOs = Qs + TD_Q * TD_RP;                                // backward: [32][TD_RP] dO
    bf16_t* Qt = dOs + TD_Q * TD_RP;                                // backward: [32 d][TD_PP] Q^T
    bf16_t* dOt = Qt + TD_HD * TD_PP;                               // backward: [32 d][TD_PP] dO^T
    bf16_t* Pt = dOt + TD_HD * TD_PP;                               // backward: [KMAX key][TD_PP] (P * dropout factor)^T
    bf16_t* dSt = Pt + KMAX * TD_PP;                                // backward: [KMAX key][TD_PP] dS^T
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const TrainAttnIdx ix = train_attn_idx(a);
    const int h = ix.h;
    const int Lq = a.Lq, Lk = a.Lk;
    const int njt = (Lk + 15) >> 4, nkk = (njt + 1) >> 1, NKP = 32 * nkk;      // key tiles, 32-key k-steps, padded key count
    const float sl2 = a.scale * 1.44269504088896340736f;
    // pass_loop (> 1, with pass_B and kv_shared): ONE workgroup per (image bl, head) walks that image's pass_loop permutation passes — K / V
    // are staged once for all of them and dK / dV accumulate in the matrix-core accumulators across the passes, one store at the end
    // (the cross-attention over the encoder memory: its K | V and their gradients are 300 MB per pass otherwise).  Else one pass: p0.
    const int npass = a.pass_loop > 1 ? a.pass_loop : 1;

    const float* kg = a.k + (size_t)ix.bkv * Lk * a.ldkv + h * TD_HD;
    const float* vg = a.v + (size_t)ix.bkv * Lk * a.ldkv + h * TD_HD;
    if constexpr (BACKWARD) {
        td_stage<true, true, KMAX>(kg, a.ldkv, Lk, NKP, Ks, XT, TP, tid);
        td_stage<true, false, KMAX>(vg, a.ldkv, Lk, NKP, Vs, nullptr, 0, tid);
    } else {
        td_stage<true, false, KMAX>(kg, a.ldkv, Lk, NKP, Ks, nullptr, 0, tid);
        td_stage<false, true, KMAX>(vg, a.ldkv, Lk, NKP, nullptr, XT, TP, tid);
    }
    f32x4 gkacc[JJ][2], gvacc[JJ][2];                                // backward: dK / dV tiles (jj, dt) of this wave, over the passes
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { gkacc[jj][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; gvacc[jj][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int pp = 0; pp < npass; ++pp) {
    // b: the image over all passes (rows of q / o / d_o / dq); with pass_loop the launch's images are bl and the passes come from the loop
    const int b = a.pass_loop > 1 ? pp * a.pass_B + ix.bl : ix.bf;
    const unsigned char* qmask = a.pass_loop > 1 ? (a.qmask ? a.qmask + (size_t)pp * a.qmask_pstride : nullptr) : ix.qmask;
    const unsigned site = a.pass_loop > 1 ? a.drop_site + (unsigned)pp * a.site_pstride : ix.site;
    const float* qg = a.q + (size_t)b * a.q_bstride + h * TD_HD;
    if (pp) __syncthreads();                     // the previous pass's readers are done with Q / dO / P^T / dS^T
    if constexpr (BACKWARD) {
        td_stage<true, true, TD_Q>(qg, a.ldq, Lq, TD_Q, Qs, Qt, TD_PP, tid);
        td_stage<true, true, TD_Q>(a.d_o + (size_t)b * Lq * a.ldo + h * TD_HD, a.ldo, Lq, TD_Q, dOs, dOt, TD_PP, tid);
    } else {
        td_stage<true, false, TD_Q>(qg, a.ldq, Lq, TD_Q, Qs, nullptr, 0, tid);
    }
    __syncthreads();

    // ---- S^T (and dP^T) of this wave's 16 queries; one 32-wide k-step over the head width
    const int l = 16 * wave + r16;                                  // this lane's query
    Frag<bf16_t> qf, of;
    qf.v = *reinterpret_cast<const bf16x8*>(Qs + l * TD_RP + 8 * g);
    if constexpr (BACKWARD) of.v = *reinterpret_cast<const bf16x8*>(dOs + l * TD_RP + 8 * g);
    f32x4 sacc[KT], pacc[KT];
    float fdrop[KT][4];                                             // backward: dropout factor of (query l, key)
    const unsigned long long drow = (ix.dblock * Lq + l) * Lk;
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < KT; ++jt) {
        sacc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BACKWARD) pacc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (jt < 2 * nkk) {
            Frag<bf16_t> kf;
            kf.v = *reinterpret_cast<const bf16x8*>(Ks + (16 * jt + r16) * TD_RP + 8 * g);
            mma16(sacc[jt], kf, qf);
            if constexpr (BACKWARD) {
                Frag<bf16_t> vf;
                vf.v = *reinterpret_cast<const bf16x8*>(Vs + (16 * jt + r16) * TD_RP + 8 * g);
                mma16(pacc[jt], vf, of);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            bool masked = j >= Lk;
            if (!masked && l < Lq) masked = (qmask && qmask[(size_t)l * Lk + j]) || (a.kmask && a.kmask[(size_t)ix.bl * a.ldkm + j]);
            if (masked) sacc[jt][r] = -INFINITY;
            mx = fmaxf(mx, sacc[jt][r]);
        }
    }
    mx = rows4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f((sacc[jt][r] - mx) * sl2); sacc[jt][r] = e; sum += e; }
    sum = rows4_sum(sum);
    const float inv = 1.0f / sum;
    if constexpr (!BACKWARD) {
        // ---- O^T = V^T (P f)^T: un-normalised probabilities times the dropout factor as the B operand, 1 / sum on the result
        Frag<bf16_t> pf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int jt = 2 * kk + (s8 >> 2), r = s8 & 3, j = 16 * jt + 4 * g + r;
                float e = sacc[jt][r];
                if (a.drop.thresh && j < Lk && l < Lq) e *= drop_factor(a.drop, site, drow + j);
                pf[kk].v[s8] = static_cast<bf16_t>(e);
            }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                if (kk < nkk) mma16(oacc, td_keyslot_frag<TP>(XT, 16 * dt + r16, kk, g), pf[kk]);
            if (l < Lq) *reinterpret_cast<f32x4*>(a.o + ((size_t)b * Lq + l) * a.ldo + h * TD_HD + 16 * dt + 4 * g) = oacc * inv;
        }
    } else {
        // ---- P = e / sum;  PD = P f (what multiplied V);  dp = dP f;  dS = P (dp - sum_j dp P) scale
        float dot = 0.f;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * g + r;
                const float f = (a.drop.thresh && j < Lk && l < Lq) ? drop_factor(a.drop, site, drow + j) : 1.0f;
                fdrop[jt][r] = f;
                sacc[jt][r] *= inv;
                pacc[jt][r] *= f;
                dot += sacc[jt][r] * pacc[jt][r];
            }
        dot = rows4_sum(dot);
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = sacc[jt][r];
                const float ds = p * (pacc[jt][r] - dot) * a.scale;
                pacc[jt][r] = ds;
                if (jt < 2 * nkk) {
                    Pt[(16 * jt + 4 * g + r) * TD_PP + l] = static_cast<bf16_t>(p * fdrop[jt][r]);
                    dSt[(16 * jt + 4 * g + r) * TD_PP + l] = static_cast<bf16_t>(ds);
                }
            }
        // ---- dQ^T = K^T dS^T, dS^T from the registers
        Frag<bf16_t> df[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) df[kk].v[s8] = static_cast<bf16_t>(pacc[2 * kk + (s8 >> 2)][s8 & 3]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 qacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                if (kk < nkk) mma16(qacc, td_keyslot_frag<TP>(XT, 16 * dt + r16, kk, g), df[kk]);
            if (l < Lq) *reinterpret_cast<f32x4*>(a.dq + ((size_t)b * Lq + l) * a.lddq + h * TD_HD + 16 * dt + 4 * g) = qacc;
        }
        __syncthreads();                         // (P f)^T and dS^T of all 32 queries are in LDS
        // ---- dV += (P f)^T dO, dK += dS^T Q over the 32 queries (one k-step): key tiles jt = wave, wave + 2, ...
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const int jt = wave + 2 * jj;
            if (jt < njt) {
                Frag<bf16_t> pa, sa;
                pa.v = *reinterpret_cast<const bf16x8*>(Pt + (16 * jt + r16) * TD_PP + 8 * g);
                sa.v = *reinterpret_cast<const bf16x8*>(dSt + (16 * jt + r16) * TD_PP + 8 * g);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    Frag<bf16_t> ob, qb;
                    ob.v = *reinterpret_cast<const bf16x8*>(dOt + (16 * dt + r16) * TD_PP + 8 * g);
                    qb.v = *reinterpret_cast<const bf16x8*>(Qt + (16 * dt + r16) * TD_PP + 8 * g);
                    // the TRANSPOSED tiles (operands swapped: the same products in the same order): a lane holds four consecutive head
                    // columns of ONE key — gv[r] = dV[key 16 jt + r16][d = 16 dt + 4 g + r] — so the read-modify-write of dK / dV is one
                    // 16-byte access per lane and tile instead of four 4-byte ones
                    mma16(gvacc[jj][dt], ob, pa);
                    mma16(gkacc[jj][dt], qb, sa);
                }
            }
        }
    }
    }      // passes
    if constexpr (BACKWARD) {
        // dK / dV leave once: rows of image ix.bf (pass_loop: ix.bl — one copy for all the passes), added to the old values if kv_accumulate
        const int bo = a.pass_loop > 1 ? ix.bl : ix.bf;
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const int jt = wave + 2 * jj, j = 16 * jt + r16;
            if (jt < njt && j < Lk) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const size_t gi = ((size_t)bo * Lk + j) * a.lddkv + h * TD_HD + 16 * dt + 4 * g;
                    f32x4 ok_ = f32x4{0.f, 0.f, 0.f, 0.f}, ov_ = f32x4{0.f, 0.f, 0.f, 0.f};      // old values (accumulate): both requested before the first store
                    if (a.kv_accumulate) { ok_ = *reinterpret_cast<const f32x4*>(a.dk + gi); ov_ = *reinterpret_cast<const f32x4*>(a.dv + gi); }
                    *reinterpret_cast<f32x4*>(a.dk + gi) = ok_ + gkacc[jj][dt];
                    *reinterpret_cast<f32x4*>(a.dv + gi) = ov_ + gvacc[jj][dt];
                }
            }
        }
    }
}

// im2col of the patch embedding: row (b, gy, gx), column (c, ky, kx) = img[b][c][gy * ph + ky][gx * pw + kx]   (fp32)
static __global__ __launch_bounds__(256)
void patches_kernel(const float* __restrict__ img, int H, int W, int ph, int pw, float* __restrict__ out) {
    const int gw = W / pw, gh = H / ph, pk = 3 * ph * pw;
    const int row = blockIdx.x, b = row / (gh * gw), gy = (row / gw) % gh, gx = row % gw;
    for (int col = threadIdx.x; col < pk; col += 256) {
        const int c = col / (ph * pw), ky = (col / pw) % ph, kx = col % pw;
        out[(size_t)row * pk + col] = img[(((size_t)b * 3 + c) * H + gy * ph + ky) * W + gx * pw + kx];
    }
}

// -------------------------------------------------------------------------------------------------------------------
// optimiser step over the flat parameter / gradient buffers (timm create_optimizer_v2('adamw') = torch.optim.AdamW;
// gradient clipping = torch.nn.utils.clip_grad_norm_, what Lightning's gradient_clip_val applies; configs/main.yaml:39)
// -------------------------------------------------------------------------------------------------------------------
constexpr int SUMSQ_BLOCKS = 1024;

// partial[block] = sum of g[i]^2 over the block's grid-stride slice (fixed order: deterministic)
static __global__ __launch_bounds__(256)
void sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)SUMSQ_BLOCKS * 256) s = fmaf(g[i], g[i], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
static __global__ __launch_bounds__(256)
void sumsq_final_kernel(const float* __restrict__ partial, float* __restrict__ norm_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < SUMSQ_BLOCKS; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *norm_out = sqrtf(red[0]);
}

// torch.optim.AdamW (single-tensor path), with the clip coefficient min(1, max_norm / (norm + 1e-6)) applied to the gradient
// on the fly when `norm` is given:   p *= 1 - lr wd;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g^2;
//                                    p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
static __global__ __launch_bounds__(256)
void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt, const float* __restrict__ norm, float max_norm) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float coef = 1.f;
    if (norm) coef = fminf(max_norm / (*norm + 1e-6f), 1.0f);
    const float grad = g[i] * coef;
    float param = p[i] * (1.0f - lr * weight_decay);
    const float mi = m[i] + (grad - m[i]) * (1.0f - beta1);
    const float vi = v[i] * beta2 + (1.0f - beta2) * grad * grad;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    param -= (lr / bc1) * (mi / denom);
    p[i] = param; m[i] = mi; v[i] = vi;
}

}  // namespace pq
