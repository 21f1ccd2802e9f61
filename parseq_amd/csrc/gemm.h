// LDS-tiled MFMA GEMM for every Linear on the PARSeq path:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue).
//
// W is the torch nn.Linear weight as-is ([out_features, in_features], K contiguous) — already the layout the MFMA
// B-operand wants, so no weight transposition anywhere.  Storage type T is bf16 (throughput mode,
// v_mfma_f32_16x16x32_bf16) or f32 (exact mode, v_mfma_f32_16x16x4_f32); accumulation is always fp32.
//
// Tile: BM x BN outputs per workgroup, 128 BYTES of K per stage (64 bf16 / 32 f32), WM x WN waves of 64 lanes,
// each wave owning (BM/WM) x (BN/WN) outputs as 16x16 MFMA tiles.  Two LDS stages; the next stage's global loads
// are issued into registers before the current stage's MFMAs and written to LDS after them (one barrier per stage).
// LDS rows are padded 128 -> 144 bytes so the 16-lane groups of ds_read_b128 fall on distinct 16-byte slots.
//
// Operand order per tile is chosen by the epilogue: "n4" form (W first) leaves each lane with 4 consecutive n for one
// m -> vector stores along a row of a row-major output; "m4" form (A first) leaves 4 consecutive m for one n ->
// vector stores into a transposed output (the V^T the encoder attention kernel wants).
//
// Workgroup id -> tile mapping is XCD-aware: ids that land on the same XCD (id % 8, observed dispatch) walk the n-tiles
// of the same m-tile back to back, so an A row-panel is fetched into one XCD's L2 once instead of up to 8 times.
#pragma once
#include "common.h"

namespace pq {

constexpr int GEMM_KB = 128;        // bytes of K per LDS stage
constexpr int GEMM_ROWB = 144;      // padded LDS row pitch in bytes

// ---------------------------------------------------------------------------------------------------------------
// A-operand loaders: produce the 16-byte chunk (row m, elements [k, k + 16/sizeof(T))) of the logical A matrix.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct ARowMajor {
    const T* A;
    int lda;
    __device__ __forceinline__ u32x4 load(int m, int k) const {
        return *reinterpret_cast<const u32x4*>(A + (size_t)m * lda + k);
    }
};

// im2col-free patch embedding: logical A[m = (b, gy, gx)][k = (c, ky, kx)] = img[b][c][gy*ph + ky][gx*pw + kx].
// One 16-byte chunk is a run of kx inside one image row (pw is a multiple of the chunk length), so it is read
// straight from the image with one (f32 source, bf16 compute: two) vector load.  Timm PatchEmbed + Conv2d weight
// [E, 3, ph, pw] flattened is exactly W[E][k] with this k order.
template <typename T, typename TI>
struct APatch {
    const TI* img;
    int C, H, Wd, ph, pw, gw, tokens;   // tokens = gh * gw
    __device__ __forceinline__ u32x4 load(int m, int k) const {
        const int b = m / tokens, t = m - b * tokens;
        const int gy = t / gw, gx = t - gy * gw;
        const int c = k / (ph * pw), r = k - c * ph * pw;
        const int ky = r / pw, kx = r - ky * pw;
        const TI* src = img + (((size_t)b * C + c) * H + (gy * ph + ky)) * Wd + gx * pw + kx;
        constexpr int n = 16 / (int)sizeof(T);
        if constexpr (sizeof(TI) == sizeof(T)) {
            return *reinterpret_cast<const u32x4*>(src);                 // same storage type: one 16-byte load
        } else if constexpr (sizeof(TI) == 4) {                          // f32 image -> bf16 operand: 2 x 16-byte loads
            const float4 lo = reinterpret_cast<const float4*>(src)[0], hi = reinterpret_cast<const float4*>(src)[1];
            union { u32x4 u; T e[n]; } out;
            out.e[0] = from_f32<T>(lo.x); out.e[1] = from_f32<T>(lo.y); out.e[2] = from_f32<T>(lo.z); out.e[3] = from_f32<T>(lo.w);
            out.e[4] = from_f32<T>(hi.x); out.e[5] = from_f32<T>(hi.y); out.e[6] = from_f32<T>(hi.z); out.e[7] = from_f32<T>(hi.w);
            return out.u;
        } else {                                                         // bf16 image -> f32 operand: one 8-byte load
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(src);
            union { u32x4 u; float e[4]; } out;
            out.e[0] = (float)v[0]; out.e[1] = (float)v[1]; out.e[2] = (float)v[2]; out.e[3] = (float)v[3];
            return out.u;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Epilogues.  n4(m, n, v): v[j] = C[m][n + j];  m4(m, n, v): v[j] = C[m + j][n].  Bounds are the epilogue's job.
// ---------------------------------------------------------------------------------------------------------------
struct EpiBase {
    int M, N;
    const float* bias;   // [N] or nullptr
    __device__ __forceinline__ bool transposed(int) const { return false; }
    __device__ __forceinline__ void m4(int, int, const float*) const {}
    __device__ __forceinline__ float b(int n) const { return bias ? bias[n] : 0.f; }
};

// out[m][n] = acc + bias (stored as TO), row-major with leading dimension ldo; optional row remap
// out_row = (m / period) * stride + offset + (m % period)  (decoder head writes step/pass rows into [B, L, C]).
template <typename TO>
struct EpiStore : EpiBase {
    TO* out; int ldo; int period, stride, offset; float scale;
    __device__ __forceinline__ void n4(int m, int n, const float* v) const {
        if (m >= M) return;
        const int row = period ? (m / period) * stride + offset + (m % period) : m;
        TO* p = out + (size_t)row * ldo + n;
        if (n + 3 < N && (ldo & 3) == 0) {
            float o[4] = {(v[0] + b(n)) * scale, (v[1] + b(n + 1)) * scale, (v[2] + b(n + 2)) * scale, (v[3] + b(n + 3)) * scale};
            store4<TO>(p, o);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < N) p[j] = from_f32<TO>((v[j] + b(n + j)) * scale);
        }
    }
};

// out[m][n] = gelu(acc + bias)
template <typename TO>
struct EpiGelu : EpiBase {
    TO* out; int ldo;
    __device__ __forceinline__ void n4(int m, int n, const float* v) const {
        if (m >= M || n + 3 >= N) return;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gelu_erf(v[j] + b(n + j));
        store4<TO>(out + (size_t)m * ldo + n, o);
    }
};

// x[m][n] += acc + bias   (fp32 residual stream, in place)
struct EpiResid : EpiBase {
    float* x; int ldx;
    __device__ __forceinline__ void n4(int m, int n, const float* v) const {
        if (m >= M || n + 3 >= N) return;
        float4* p = reinterpret_cast<float4*>(x + (size_t)m * ldx + n);
        float4 r = *p;
        r.x += v[0] + b(n); r.y += v[1] + b(n + 1); r.z += v[2] + b(n + 2); r.w += v[3] + b(n + 3);
        *p = r;
    }
};

// x[m][n] = acc + bias + table[(m % period) + offset][n]   (patch embed + pos_embed; decoder query + pos_queries)
struct EpiAddTable : EpiBase {
    float* x; int ldx; const float* table; int ldt, period, offset;
    __device__ __forceinline__ void n4(int m, int n, const float* v) const {
        if (m >= M || n + 3 >= N) return;
        const float4 t = *reinterpret_cast<const float4*>(table + (size_t)((m % period) + offset) * ldt + n);
        float o[4] = {v[0] + b(n) + t.x, v[1] + b(n + 1) + t.y, v[2] + b(n + 2) + t.z, v[3] + b(n + 3) + t.w};
        store4<float>(x + (size_t)m * ldx + n, o);
    }
};

// Fused encoder qkv projection output: columns [0,E) -> q[b][h][t][d], [E,2E) -> k[b][h][t][d] (row-major per head),
// [2E,3E) -> vt[b][h][d][t] (transposed per head; written from the m4 form so the 4 values are 4 consecutive tokens).
template <typename TO>
struct EpiQKV : EpiBase {
    TO *q, *k, *vt; int E, heads, hd, tokens;
    __device__ __forceinline__ bool transposed(int n0) const { return n0 >= 2 * E; }
    __device__ __forceinline__ void n4(int m, int n, const float* v) const {
        if (m >= M || n + 3 >= N) return;
        const int which = n / E, c = n - which * E;      // 4 consecutive n never straddle a head (hd % 4 == 0)
        const int h = c / hd, d = c - h * hd;
        const int b_ = m / tokens, t = m - b_ * tokens;
        TO* dst = (which == 0 ? q : k) + (((size_t)b_ * heads + h) * tokens + t) * hd + d;
        float o[4] = {v[0] + b(n), v[1] + b(n + 1), v[2] + b(n + 2), v[3] + b(n + 3)};
        store4<TO>(dst, o);
    }
    __device__ __forceinline__ void m4(int m, int n, const float* v) const {
        if (m + 3 >= M || n >= N) return;                // M is a multiple of tokens (a multiple of 4)
        const int c = n - 2 * E;
        const int h = c / hd, d = c - h * hd;
        const int b_ = m / tokens, t = m - b_ * tokens;
        const float bb = b(n);
        float o[4] = {v[0] + bb, v[1] + bb, v[2] + bb, v[3] + bb};
        store4<TO>(vt + (((size_t)b_ * heads + h) * hd + d) * tokens + t, o);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, typename ALoad, typename Epi>
__global__ __launch_bounds__(WM * WN * 64)
void gemm_kernel(const ALoad aload, const T* __restrict__ W, int ldw, int M, int N, int K, int mtiles, int ntiles,
                 const Epi epi) {
    constexpr int NT = WM * WN * 64;
    constexpr int EPC = 16 / (int)sizeof(T);          // elements per 16-byte chunk
    constexpr int BK = GEMM_KB / (int)sizeof(T);      // elements of K per stage
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_IT = BM * 8 / NT, W_IT = BN * 8 / NT;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/thread mismatch");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                               // [2][BM][ROWB]
    unsigned char* Ws = smem + 2 * BM * GEMM_ROWB;          // [2][BN][ROWB]

    // XCD-aware tile mapping (bijective over a grid rounded up to 8 * ceil(mtiles / 8) * ntiles)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = slot % ntiles, mt = (slot / ntiles) * 8 + xcd;
    if (mt >= mtiles) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const bool tr = epi.transposed(n0);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Staging: each thread owns A_IT + W_IT 16-byte chunks of a stage.  Loads are unconditional from clamped addresses
    // (no divergent control flow, everything stays in registers); out-of-range chunks are zeroed by a select.
    u32x4 ra[A_IT], rw[W_IT];
    int a_row[A_IT], w_row[W_IT];
    bool a_ok[A_IT], w_ok[W_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int r = m0 + ((it * NT + tid) >> 3);
        a_ok[it] = r < M; a_row[it] = a_ok[it] ? r : M - 1;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int r = n0 + ((it * NT + tid) >> 3);
        w_ok[it] = r < N; w_row[it] = w_ok[it] ? r : N - 1;
    }
    const int kc = (tid & 7) * EPC;            // NT is a multiple of 8, so the chunk column is the same for every `it`
    const int klast = K - EPC;
    const u32x4 zero = {0u, 0u, 0u, 0u};

#define PQ_GLOAD(k0)                                                                                  \
    {                                                                                                 \
        const int k_ = (k0) + kc;                                                                     \
        const bool kin_ = k_ < K;                                                                     \
        const int kk_ = kin_ ? k_ : klast;                                                            \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                         \
            const u32x4 v_ = aload.load(a_row[it], kk_);                                              \
            ra[it] = (a_ok[it] && kin_) ? v_ : zero;                                                  \
        }                                                                                             \
        _Pragma("unroll") for (int it = 0; it < W_IT; ++it) {                                         \
            const u32x4 v_ = *reinterpret_cast<const u32x4*>(W + (size_t)w_row[it] * ldw + kk_);      \
            rw[it] = (w_ok[it] && kin_) ? v_ : zero;                                                  \
        }                                                                                             \
    }
#define PQ_LSTORE(buf)                                                                                                 \
    {                                                                                                                  \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                                          \
            const int c_ = it * NT + tid;                                                                              \
            *reinterpret_cast<u32x4*>(As + ((buf) * BM + (c_ >> 3)) * GEMM_ROWB + (c_ & 7) * 16) = ra[it];             \
        }                                                                                                              \
        _Pragma("unroll") for (int it = 0; it < W_IT; ++it) {                                                          \
            const int c_ = it * NT + tid;                                                                              \
            *reinterpret_cast<u32x4*>(Ws + ((buf) * BN + (c_ >> 3)) * GEMM_ROWB + (c_ & 7) * 16) = rw[it];             \
        }                                                                                                              \
    }

    const int nk = (K + BK - 1) / BK;
    PQ_GLOAD(0)
    PQ_LSTORE(0)
    __syncthreads();

    // First MFMA operand P supplies the output's register dimension (4 consecutive indices per lane), second operand Q
    // the lane dimension.  n4 form: P = W rows (n), Q = A rows (m).  m4 form (transposed stores): P = A, Q = W.
    // Wave tiles are square (static_assert below), so swapping the roles is just swapping two LDS base pointers.
    static_assert(BM / WM == BN / WN && TM == TN, "square wave tiles required for the operand-role swap");
    const int frow = lane & 15, fk = (lane >> 4) * 16;
    const int a_off = (wm * (BM / WM) + frow) * GEMM_ROWB + fk;
    const int w_off = (wn * (BN / WN) + frow) * GEMM_ROWB + fk;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) PQ_GLOAD((kt + 1) * BK)
        const unsigned char* Ab = As + cur * BM * GEMM_ROWB + a_off;
        const unsigned char* Wb = Ws + cur * BN * GEMM_ROWB + w_off;
        const unsigned char* Pb = tr ? Ab : Wb;
        const unsigned char* Qb = tr ? Wb : Ab;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            Frag<T> fp[TM], fq[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fp[i].v = *reinterpret_cast<const decltype(fp[i].v)*>(Pb + i * 16 * GEMM_ROWB + kk * 64);
#pragma unroll
            for (int j = 0; j < TN; ++j) fq[j].v = *reinterpret_cast<const decltype(fq[j].v)*>(Qb + j * 16 * GEMM_ROWB + kk * 64);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma16(acc[i][j], fp[i], fq[j]);
        }
        if (kt + 1 < nk) PQ_LSTORE(cur ^ 1)
        __syncthreads();
    }

#undef PQ_GLOAD
#undef PQ_LSTORE
    const int mb = m0 + wm * (BM / WM), nb = n0 + wn * (BN / WN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (!tr) epi.n4(mb + j * 16 + (lane & 15), nb + i * 16 + 4 * (lane >> 4), v);
            else     epi.m4(mb + i * 16 + 4 * (lane >> 4), nb + j * 16 + (lane & 15), v);
        }
}

template <int BM, int BN>
constexpr size_t gemm_lds_bytes() { return (size_t)2 * (BM + BN) * GEMM_ROWB; }

template <typename T, int BM, int BN, int WM, int WN, typename ALoad, typename Epi>
inline hipError_t launch_gemm(hipStream_t s, const ALoad& aload, const T* W, int ldw, int M, int N, int K, const Epi& epi) {
    const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
    const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
    auto kern = gemm_kernel<T, BM, BN, WM, WN, ALoad, Epi>;
    constexpr size_t lds = gemm_lds_bytes<BM, BN>();
    if (lds > 64 * 1024) {
        static bool attr_done = false;      // one flag per template instantiation
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, s, aload, W, ldw, M, N, K, mtiles, ntiles, epi);
    return hipGetLastError();
}

}  // namespace pq
