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
// SPLIT (precision bf16x3, T = float): every operand value v is carried as a PAIR of bf16, hi = bf16(v) and lo = bf16(v - hi)
// (16 mantissa bits between them), and a product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation: ~2^-17 relative error per product instead of bf16's 2^-9, at 3/16 of the exact-f32 MFMA's cycles per
// product.  Layout "block-planar": 32 consecutive elements of a row occupy 128 bytes — 64 bytes of hi then 64 bytes of lo —
// so a row of K elements is still K * 4 bytes and all tile / stage address arithmetic is shared with f32.  Weights are stored
// that way once (split_pack_kernel); activations arrive as plain f32 and are split by the A-loader's staging pass (once per
// tile load, not once per MFMA).  A fragment of the 32-deep k-group is then 16 bytes of the hi plane and 16 bytes of the lo
// plane at the same offset: the main loop is three MFMAs per tile pair and no conversion.
//
// Workgroup id -> tile mapping is XCD-aware: ids that land on the same XCD (id % 8, observed dispatch) walk the n-tiles
// of the same m-tile back to back, so an A row-panel is fetched into one XCD's L2 once instead of up to 8 times.
#pragma once
#include <cstdlib>

#include "common.h"

// (Removed, kept in the history at commit c342b0c, the parent of the pruning commit c0ffb4f: the four-stage counted-vmcnt ring of the pre-split (PAIRS) loop — one workgroup
// per CU, measured slower than two buffers x two workgroups: qkv 254 -> 305 us, fc1 347 -> 430, fc2 233 -> 261 — and the diagnostic
// builds of the bf16x3 co-residency defect hunt, tools/x3_diag2.py.)

namespace pq {

// Stage geometry.  KB = bytes of K per LDS stage: 128 for the encoder's M = batch*128 GEMMs (double-buffered), 768 for
// the decoder's small-M GEMMs (K = 384 bf16 in ONE stage: one exposed load latency per GEMM instead of six; single
// buffer).  LDS rows are padded by 16 bytes so the 16-lane groups of ds_read_b128 fall on distinct 16-byte slots.
template <int KB> constexpr int gemm_rowb() { return KB + 16; }

// ---------------------------------------------------------------------------------------------------------------
// A-operand loaders: produce the 16-byte chunk (row m, elements [k, k + 16/sizeof(T))) of the logical A matrix.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct ARowMajor {
    const T* A;
    int lda;
    static constexpr int kStatsFloats = 0;       // LDS floats per tile row the loader needs (see ALayerNorm)
    static constexpr bool kDirect = true;        // plain row-major T: eligible for the global_load_lds main loop
    struct Raw { u32x4 v; };
    __device__ __forceinline__ void prepare(int, int, int, float*) {}
    // fetch() only ISSUES the global loads; finish() turns the landed bytes into the 16-byte LDS chunk.  The kernel calls
    // finish() after the current stage's MFMAs, so nothing between the loads and the MFMAs depends on the loaded data.
    __device__ __forceinline__ Raw fetch(int m, int k) const { return Raw{*reinterpret_cast<const u32x4*>(A + (size_t)m * lda + k)}; }
    __device__ __forceinline__ u32x4 finish(const Raw& r, int, int) const { return r.v; }
};

// im2col-free patch embedding: logical A[m = (b, gy, gx)][k = (c, ky, kx)] = img[b][c][gy*ph + ky][gx*pw + kx].
// One 16-byte chunk is a run of kx inside one image row (pw is a multiple of the chunk length), so it is read
// straight from the image with one (f32 source, bf16 compute: two) vector load.  Timm PatchEmbed + Conv2d weight
// [E, 3, ph, pw] flattened is exactly W[E][k] with this k order.
template <typename T, typename TI>
struct APatch {
    const TI* img;
    int C, H, Wd, ph, pw, gw, tokens;   // tokens = gh * gw
    static constexpr int kStatsFloats = 0;
    static constexpr bool kDirect = false;
    static constexpr int kRaw = (sizeof(TI) > sizeof(T)) ? 2 : 1;       // 16-byte loads per chunk
    static constexpr bool kU8 = sizeof(TI) == 1;                        // raw pixels: ToTensor + Normalize fused here
    struct Raw { u32x4 v[kRaw]; };
    // strhub/data/module.py:78-81: ToTensor (v / 255 in f32) then Normalize(0.5, 0.5) ((t - 0.5) / 0.5), IEEE-exact
    static __device__ __forceinline__ float norm_u8(unsigned v) { return ((float)v / 255.0f - 0.5f) / 0.5f; }
    __device__ __forceinline__ void prepare(int, int, int, float*) {}
    __device__ __forceinline__ Raw fetch(int m, int k) const {
        const int b = m / tokens, t = m - b * tokens;
        const int gy = t / gw, gx = t - gy * gw;
        const int c = k / (ph * pw), r = k - c * ph * pw;
        const int ky = r / pw, kx = r - ky * pw;
        const TI* src = img + (((size_t)b * C + c) * H + (gy * ph + ky)) * Wd + gx * pw + kx;
        Raw raw;
        if constexpr (kU8) {                                                       // 16 / sizeof(T) pixels of one image row
            if constexpr (sizeof(T) == 2) { const uint2 b8 = *reinterpret_cast<const uint2*>(src); raw.v[0] = u32x4{b8.x, b8.y, 0u, 0u}; }
            else raw.v[0] = u32x4{*reinterpret_cast<const unsigned*>(src), 0u, 0u, 0u};
        } else if constexpr (sizeof(TI) == sizeof(T)) {
            raw.v[0] = *reinterpret_cast<const u32x4*>(src);                       // same storage type
        } else if constexpr (sizeof(TI) == 4) {                                    // f32 image -> bf16 operand: 8 floats
            raw.v[0] = reinterpret_cast<const u32x4*>(src)[0];
            raw.v[1] = reinterpret_cast<const u32x4*>(src)[1];
        } else {                                                                   // bf16 image -> f32 operand: 4 bf16
            const uint2 h2 = *reinterpret_cast<const uint2*>(src);
            raw.v[0] = u32x4{h2.x, h2.y, 0u, 0u};
        }
        return raw;
    }
    __device__ __forceinline__ u32x4 finish(const Raw& raw, int, int) const {
        if constexpr (kU8) {
            constexpr int NE = 16 / (int)sizeof(T);
            union { u32x4 u; T e[NE]; } out;
#pragma unroll
            for (int i = 0; i < NE; ++i) out.e[i] = from_f32<T>(norm_u8((raw.v[0][i / 4] >> (8 * (i % 4))) & 0xffu));
            return out.u;
        } else if constexpr (sizeof(TI) == sizeof(T)) {
            return raw.v[0];
        } else if constexpr (sizeof(TI) == 4) {
            union { u32x4 u; T e[8]; } out;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out.e[i] = from_f32<T>(__uint_as_float(raw.v[0][i]));
                out.e[4 + i] = from_f32<T>(__uint_as_float(raw.v[1][i]));
            }
            return out.u;
        } else {
            u32x4 out;      // bf16 -> f32 is a 16-bit shift
            out[0] = raw.v[0][0] << 16; out[1] = raw.v[0][0] & 0xffff0000u;
            out[2] = raw.v[0][1] << 16; out[3] = raw.v[0][1] & 0xffff0000u;
            return out;
        }
    }
};

// (x - mean) * rstd * gamma + beta for four elements, every intermediate pinned to a scalar-f32 register (the empty asm keeps the
// SLP vectoriser from fusing neighbours into v_pk_mul_f32 / v_pk_fma_f32).  Not a micro-optimisation: with packed f32 arithmetic
// here, the bf16x3 form of the 128 x 128-tile GEMM (whose LDS store path continues with v_cvt_pk_bf16_f32 / subtract / convert
// on the same registers) produced wrong values in lanes 48-63 of the staging pass — rows 6, 7 mod 8 of a tile, some k-chunks of
// them — non-deterministically and only while two workgroups shared a compute unit.  Draining VMEM before the stores, unmerged
// 8-byte LDS stores and wait states between the converts and the stores all left it in place; this form is exact and
// deterministic on every shape tools/x3_diag2.py runs (the cause at the instruction level is not identified).
__device__ __forceinline__ void ln_apply4(const u32x4& raw, float mean, float rstd, const float4& gv, const float4& bv, float (&o)[4]) {
    float d0 = __uint_as_float(raw[0]) - mean, d1 = __uint_as_float(raw[1]) - mean, d2 = __uint_as_float(raw[2]) - mean, d3 = __uint_as_float(raw[3]) - mean;
    asm volatile("" : "+v"(d0)); asm volatile("" : "+v"(d1)); asm volatile("" : "+v"(d2)); asm volatile("" : "+v"(d3));
    d0 *= rstd; asm volatile("" : "+v"(d0)); d1 *= rstd; asm volatile("" : "+v"(d1));
    d2 *= rstd; asm volatile("" : "+v"(d2)); d3 *= rstd; asm volatile("" : "+v"(d3));
    d0 = d0 * gv.x + bv.x; asm volatile("" : "+v"(d0)); d1 = d1 * gv.y + bv.y; asm volatile("" : "+v"(d1));
    d2 = d2 * gv.z + bv.z; asm volatile("" : "+v"(d2)); d3 = d3 * gv.w + bv.w; asm volatile("" : "+v"(d3));
    o[0] = d0; o[1] = d1; o[2] = d2; o[3] = d3;
}

// LayerNorm fused into the A operand: logical A[m][k] = LayerNorm(x[m])[k] rounded to T, with x the fp32 residual
// stream [M, E].  prepare() computes mean / rstd of the tile's rows (wave per row, two-pass, exactly like
// layernorm_kernel) into LDS; load() normalises on the fly.  Removes a kernel boundary and the normalised-activation
// round trip through HBM in front of every decoder GEMM that follows a LayerNorm.
template <typename T, int E>
struct ALayerNorm {
    const float* x; const float* gamma; const float* beta; float eps;
    int m0_; const float* stats_;
    static constexpr int kStatsFloats = 2;
    static constexpr bool kDirect = false;
    __device__ __forceinline__ void prepare(int m0, int bm, int M, float* stats) {
        m0_ = m0; stats_ = stats;
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
        constexpr int RB = 8;                       // rows in flight per wave: RB independent load/reduce chains
        for (int r0 = wid * RB; r0 < bm; r0 += nw * RB) {
            float v[RB][E / 64];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const float* xr = x + (size_t)min(m0 + r0 + j, M - 1) * E;
#pragma unroll
                for (int i = 0; i < E / 64; ++i) v[j][i] = xr[i * 64 + lane];
            }
            // every lane ends up with the row's mean / rstd (wave_sum broadcasts): lanes 0 .. RB-1 each keep one row's pair and
            // store it — one unpredicated-by-row store for the RB rows instead of RB lane-0 stores under a saved exec mask
            float my_mean = 0.f, my_rstd = 0.f;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < E / 64; ++i) s += v[j][i];
                const float mean = wave_sum(s) * (1.0f / E);
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < E / 64; ++i) { const float d = v[j][i] - mean; ss += d * d; }
                const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / E) + eps);
                if (lane == j) { my_mean = mean; my_rstd = rstd; }
            }
            if (lane < RB && r0 + lane < bm) { stats[2 * (r0 + lane)] = my_mean; stats[2 * (r0 + lane) + 1] = my_rstd; }
        }
    }
    static constexpr int kRaw = 4 / (int)sizeof(T);                 // fp32 source: 2 loads for a bf16 chunk, 1 for f32
    struct Raw { u32x4 v[kRaw]; };
    __device__ __forceinline__ Raw fetch(int m, int k) const {
        Raw raw;
#pragma unroll
        for (int i = 0; i < kRaw; ++i) raw.v[i] = reinterpret_cast<const u32x4*>(x + (size_t)m * E + k)[i];
        return raw;
    }
    __device__ __forceinline__ u32x4 finish(const Raw& raw, int m, int k) const {
        constexpr int n = 16 / (int)sizeof(T);
        const float mean = stats_[2 * (m - m0_)], rstd = stats_[2 * (m - m0_) + 1];
        union { u32x4 u; T e[n]; } out;
#pragma unroll
        for (int i = 0; i < kRaw; ++i) {
            const float4 gv = *reinterpret_cast<const float4*>(gamma + k + 4 * i);
            const float4 bv = *reinterpret_cast<const float4*>(beta + k + 4 * i);
            float o[4];
            ln_apply4(raw.v[i], mean, rstd, gv, bv, o);
            out.e[4 * i + 0] = from_f32<T>(o[0]); out.e[4 * i + 1] = from_f32<T>(o[1]); out.e[4 * i + 2] = from_f32<T>(o[2]); out.e[4 * i + 3] = from_f32<T>(o[3]);
        }
        return out.u;
    }
};

// The same logical A with the row statistics precomputed in global memory (rowops.h ln_stats_kernel: stats[2 m] = mean,
// stats[2 m + 1] = rstd): no LDS statistics, no prologue, no barrier before the main loop.  Used by the bf16x3 big-tile GEMMs.
template <typename T, int E>
struct ALayerNormStats {
    const float* x; const float* gamma; const float* beta; const float* stats;
    static constexpr int kStatsFloats = 0;
    static constexpr bool kDirect = false;
    __device__ __forceinline__ void prepare(int, int, int, float*) {}
    static constexpr int kRaw = 4 / (int)sizeof(T);
    struct Raw { u32x4 v[kRaw]; };
    __device__ __forceinline__ Raw fetch(int m, int k) const {
        Raw raw;
#pragma unroll
        for (int i = 0; i < kRaw; ++i) raw.v[i] = reinterpret_cast<const u32x4*>(x + (size_t)m * E + k)[i];
        return raw;
    }
    __device__ __forceinline__ u32x4 finish(const Raw& raw, int m, int k) const {
        constexpr int n = 16 / (int)sizeof(T);
        const float2 st = *reinterpret_cast<const float2*>(stats + 2 * (size_t)m);
        const float mean = st.x, rstd = st.y;
        union { u32x4 u; T e[n]; } out;
#pragma unroll
        for (int i = 0; i < kRaw; ++i) {
            const float4 gv = *reinterpret_cast<const float4*>(gamma + k + 4 * i);
            const float4 bv = *reinterpret_cast<const float4*>(beta + k + 4 * i);
            float o[4];
            ln_apply4(raw.v[i], mean, rstd, gv, bv, o);
            out.e[4 * i + 0] = from_f32<T>(o[0]); out.e[4 * i + 1] = from_f32<T>(o[1]); out.e[4 * i + 2] = from_f32<T>(o[2]); out.e[4 * i + 3] = from_f32<T>(o[3]);
        }
        return out.u;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Epilogues.  Two phases (the stores were the bottleneck of the first version: 8-byte-per-lane scattered stores ran the
// encoder GEMMs store-issue-bound at ~1 TB/s; tools/gemm_bench.py, profiles/r01_gemm_sweep.md):
//   phase 1  xform_n4(m, n, v) / xform_m4: element-wise work on the accumulators in registers (bias, GELU, scale);
//            the kernel then parks the tile in LDS in the staging type S (the output's storage type);
//   phase 2  store_n(m, n, chunk) / store_m: one 16-byte chunk of S per call, chunks walked row-major by consecutive
//            threads, so global stores (and the residual stream's read-modify-write) are full-line coalesced.
// n-form: chunk = C[m][n .. n+CH);  m-form (transposed tiles): chunk = C[m .. m+CH)[n];  CH = 16 / sizeof(S).
// ---------------------------------------------------------------------------------------------------------------
struct EpiBase {
    int M, N;
    const float* bias;   // [N] or nullptr
    static constexpr bool kPrefetch = false;     // true: the kernel calls load_n() for every chunk BEFORE the staging pass
    __device__ __forceinline__ u32x4 load_n(int, int) const { return u32x4{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ bool transposed(int) const { return false; }
    __device__ __forceinline__ float b(int n) const { return (bias && n < N) ? bias[n] : 0.f; }
    __device__ __forceinline__ void xform_n4(int, int n, float* v) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b(n + j);
    }
    __device__ __forceinline__ void xform_m4(int, int n, float* v) const {
        const float bb = b(n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += bb;
    }
};

template <typename S> struct Chunk { static constexpr int CH = 16 / (int)sizeof(S); };

// out[row(m)][n] = (acc + bias) * scale, row-major with leading dimension ldo; optional row remap
// row = (m / period) * stride + offset + (m % period)  (decoder head writes step/pass rows into [B, L, C]).
template <typename TO>
struct EpiStore : EpiBase {
    using S = TO;
    TO* out; int ldo; int period, stride, offset; float scale;
    __device__ __forceinline__ void xform_n4(int m, int n, float* v) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (v[j] + b(n + j)) * scale;
    }
    __device__ __forceinline__ void store_n(int m, int n, const S* c) const {
        constexpr int CH = Chunk<S>::CH;
        const int row = period ? (m / period) * stride + offset + (m % period) : m;
        TO* p = out + (size_t)row * ldo + n;
        if (n + CH <= N && (ldo % CH) == 0) {
            *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(c);
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) if (n + j < N) p[j] = c[j];
        }
    }
    __device__ __forceinline__ void store_m(int, int, const S*) const {}
};

// out[m][n] = gelu(acc + bias)
template <typename TO>
struct EpiGelu : EpiBase {
    using S = TO;
    TO* out; int ldo;
    __device__ __forceinline__ void xform_n4(int m, int n, float* v) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_for<TO>(v[j] + b(n + j));
    }
    __device__ __forceinline__ void store_n(int m, int n, const S* c) const {
        if (n + Chunk<S>::CH <= N) *reinterpret_cast<u32x4*>(out + (size_t)m * ldo + n) = *reinterpret_cast<const u32x4*>(c);
    }
    __device__ __forceinline__ void store_m(int, int, const S*) const {}
};

// out = gelu(acc + bias) written as block-planar hi | lo bf16 pairs (the A operand of a PAIRS GEMM): element n of row m lives in
// block n / 32 of the row — 128 bytes: hi of the block's 32 elements, then lo — so a 4-element chunk is 8 bytes in each half.
struct EpiGeluSplit : EpiBase {
    using S = float;
    unsigned char* out; int ldo;            // ldo: row pitch in logical (f32-sized) elements
    __device__ __forceinline__ void xform_n4(int m, int n, float* v) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_for<float>(v[j] + b(n + j));
    }
    __device__ __forceinline__ void store_n(int m, int n, const S* c) const {
        if (n + 4 > N) return;
        union { uint2 u; bf16_t e[4]; } h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h.e[i] = static_cast<bf16_t>(c[i]);
            l.e[i] = static_cast<bf16_t>(c[i] - static_cast<float>(h.e[i]));
        }
        unsigned char* d = out + ((size_t)m * ldo + (n & ~31)) * 4 + ((n & 31) >> 2) * 8;
        *reinterpret_cast<uint2*>(d) = h.u;
        *reinterpret_cast<uint2*>(d + 64) = l.u;
    }
    __device__ __forceinline__ void store_m(int, int, const S*) const {}
};

// x[m][n] += acc + bias   (fp32 residual stream, in place).  The old x values are fetched for ALL of a thread's chunks
// at the start of the epilogue (kPrefetch), so their latency hides under the LDS staging pass instead of being paid
// once per chunk in a load -> add -> store chain (that chain ran proj / fc2 at 2.7 TB/s).
struct EpiResid : EpiBase {
    using S = float;
    static constexpr bool kPrefetch = true;
    float* x; int ldx;
    __device__ __forceinline__ u32x4 load_n(int m, int n) const {
        return *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + n);
    }
    __device__ __forceinline__ void store_n(int m, int n, const S* c, const u32x4& old) const {
        if (n + 4 > N) return;
        float4 r;
        r.x = __uint_as_float(old[0]) + c[0]; r.y = __uint_as_float(old[1]) + c[1];
        r.z = __uint_as_float(old[2]) + c[2]; r.w = __uint_as_float(old[3]) + c[3];
        *reinterpret_cast<float4*>(x + (size_t)m * ldx + n) = r;
    }
    __device__ __forceinline__ void store_m(int, int, const S*) const {}
};

// x[m][n] = acc + bias + table[(m % period) + offset][n]   (patch embed + pos_embed; decoder query + pos_queries)
struct EpiAddTable : EpiBase {
    using S = float;
    float* x; int ldx; const float* table; int ldt, period, offset;
    __device__ __forceinline__ void store_n(int m, int n, const S* c) const {
        if (n + 4 > N) return;
        const float4 t = *reinterpret_cast<const float4*>(table + (size_t)((m % period) + offset) * ldt + n);
        *reinterpret_cast<float4*>(x + (size_t)m * ldx + n) = make_float4(c[0] + t.x, c[1] + t.y, c[2] + t.z, c[3] + t.w);
    }
    __device__ __forceinline__ void store_m(int, int, const S*) const {}
};

// Head-split projection output: N = nseg * E columns, segment s = n / E goes to seg[s] as [b][h][t][d] (row-major per
// head) for s < tr_from and as [b][h][d][t] (transposed per head; those tiles run the m-form so that a chunk is CH
// consecutive tokens) for s >= tr_from.  Encoder qkv: seg = {q, k, v^T}, tr_from = 2, hd = 64.  Decoder memory K/V:
// seg = {k, v^T}, tr_from = 1, hd = 32.  hd and tokens are multiples of CH, so a chunk never straddles a head / image.
template <typename TO>
struct EpiHeads : EpiBase {
    using S = TO;
    TO* seg[3]; int E, heads, hd, tokens, tr_from;
    int m_off = 0;                 // global row of the GEMM's row 0 (a call on the tail rows of a larger activation matrix)
    __device__ __forceinline__ bool transposed(int n0) const { return n0 >= tr_from * E; }
    __device__ __forceinline__ void store_n(int m, int n, const S* c) const {
        if (n + Chunk<S>::CH > N) return;
        const int which = n / E, col = n - which * E;
        const int h = col / hd, d = col - h * hd;
        const int b_ = (m + m_off) / tokens, t = (m + m_off) - b_ * tokens;
        *reinterpret_cast<u32x4*>(seg[which] + (((size_t)b_ * heads + h) * tokens + t) * hd + d) = *reinterpret_cast<const u32x4*>(c);
    }
    __device__ __forceinline__ void store_m(int m, int n, const S* c) const {
        if (m + Chunk<S>::CH > M || n >= N) return;
        const int which = n / E, col = n - which * E;
        const int h = col / hd, d = col - h * hd;
        const int b_ = (m + m_off) / tokens, t = (m + m_off) - b_ * tokens;
        *reinterpret_cast<u32x4*>(seg[which] + (((size_t)b_ * heads + h) * hd + d) * tokens + t) = *reinterpret_cast<const u32x4*>(c);
    }
};

// Measurement aid (tools/gemm_bench.py): runs phase 1 and the LDS staging but stores nothing -> isolates the store cost.
struct EpiNull : EpiBase {
    using S = float;
    float* sink;
    __device__ __forceinline__ void store_n(int m, int n, const S* c) const { if (m < 0) sink[n] = c[0]; }
    __device__ __forceinline__ void store_m(int, int, const S*) const {}
};

// ---------------------------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------------------------
// one f32 chunk (4 values) -> 8 bytes of the hi plane + 8 bytes of the lo plane
__device__ __forceinline__ void split4(const u32x4& v, uint2& hi, uint2& lo) {
    union { uint2 u; bf16_t e[4]; } h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f = __uint_as_float(v[i]);
        h.e[i] = static_cast<bf16_t>(f);                                   // round to nearest even
        l.e[i] = static_cast<bf16_t>(f - static_cast<float>(h.e[i]));      // exact residual, rounded once
    }
    hi = h.u; lo = l.u;
}

template <typename T, int BM, int BN, int WM, int WN, int KB, int NBUF, bool DIRECT, typename ALoad, typename Epi, bool SPLIT = false, bool PAIRS = false>
__global__ __launch_bounds__(WM * WN * 64)
void gemm_kernel(const ALoad aload_, const T* __restrict__ W, int ldw, int M, int N, int K, int mtiles, int ntiles,
                 const Epi epi) {
    constexpr int NT = WM * WN * 64;
    constexpr int EPC = 16 / (int)sizeof(T);          // elements per 16-byte chunk
    constexpr int BK = KB / (int)sizeof(T);           // elements of K per stage
    constexpr int CPR = KB / 16;                      // 16-byte chunks per tile row
    constexpr int GEMM_ROWB = DIRECT ? KB : gemm_rowb<KB>();   // the LDS-DMA image is lane-linear: no padding, XOR swizzle
    static_assert(!DIRECT || (KB == 128 && NBUF == 2 && ALoad::kDirect), "direct-to-LDS loop: 128-byte stages, two buffers");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_IT = BM * CPR / NT, W_IT = BN * CPR / NT;
    static_assert(BM * CPR % NT == 0 && BN * CPR % NT == 0, "tile/thread mismatch");
    static_assert(NBUF == 1 || NBUF == 2, "one or two LDS stages");
    static_assert(!SPLIT || (sizeof(T) == 4 && !DIRECT && KB % 128 == 0), "split-bf16 products: f32 storage, register-staged loop, whole 32-element blocks");
    // PAIRS: BOTH operands already hold block-planar hi | lo bf16 pairs in memory (a 32-element block of the logical f32 matrix is 64 B
    // of hi + 64 B of lo = one 128-byte stage row), seen here as bf16 matrices of twice the logical K.  The direct-to-LDS loop moves
    // them untouched; the two 64-byte halves of a stage row are then the hi and the lo fragments of ONE k-step instead of two k-steps.
    static_assert(!PAIRS || (DIRECT && sizeof(T) == 2 && !SPLIT), "pre-split operands: bf16 view, direct-to-LDS loop");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                                  // [NBUF][BM][ROWB]
    unsigned char* Ws = smem + NBUF * BM * GEMM_ROWB;          // [NBUF][BN][ROWB]
    float* stats = reinterpret_cast<float*>(smem + NBUF * (BM + BN) * GEMM_ROWB);   // [BM][kStatsFloats]

    // XCD-aware tile mapping (bijective over a grid rounded up to 8 * ceil(mtiles / 8) * ntiles)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int nt = slot % ntiles, mt = (slot / ntiles) * 8 + xcd;
    if (mt >= mtiles) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const bool tr = epi.transposed(n0);
    ALoad aload = aload_;
    if constexpr (ALoad::kStatsFloats > 0) {
        aload.prepare(m0, BM, M, stats);
        __syncthreads();
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // First MFMA operand P supplies the output's register dimension (4 consecutive indices per lane), second operand Q
    // the lane dimension.  n4 form: P = W rows (n), Q = A rows (m).  m4 form (transposed stores): P = A, Q = W.
    // Wave tiles are square, so swapping the roles is just swapping two LDS base pointers.
    static_assert(BM / WM == BN / WN && TM == TN, "square wave tiles required for the operand-role swap");
    const int frow = lane & 15;
    const int nk = (K + BK - 1) / BK;

    if constexpr (DIRECT) {
        // ---- direct-to-LDS main loop (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass) ------------------
        // One wave instruction moves 8 tile rows x 128 bytes = 1 KiB: lane l -> LDS slot (row l >> 3, 16-byte slot l & 7)
        // of the group, fed from source chunk (l & 7) ^ (row & 7) of that row — the XOR swizzle lives on the SOURCE
        // address because the LDS destination of the DMA is lane-linear; fragment reads apply the same XOR.
        constexpr int NW = WM * WN;
        constexpr int A_LI = BM / 8 / NW, W_LI = BN / 8 / NW;
        static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row groups per wave");
        const int wu = __builtin_amdgcn_readfirstlane(wid);
        const int lrow = lane >> 3, lsrc = ((lane & 7) ^ lrow) * EPC;
        const T* a_src[A_LI]; const T* w_src[W_LI];
#pragma unroll
        for (int it = 0; it < A_LI; ++it) {
            const int r = m0 + (wu * A_LI + it) * 8 + lrow;
            a_src[it] = aload_.A + (size_t)(r < M ? r : M - 1) * aload_.lda + lsrc;
        }
#pragma unroll
        for (int it = 0; it < W_LI; ++it) {
            const int r = n0 + (wu * W_LI + it) * 8 + lrow;
            w_src[it] = W + (size_t)(r < N ? r : N - 1) * ldw + lsrc;
        }
#define PQ_DLOAD(buf, k0)                                                                                                  \
        {                                                                                                                  \
            _Pragma("unroll") for (int it = 0; it < A_LI; ++it)                                                            \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[it] + (k0)),        \
                    (__attribute__((address_space(3))) void*)(As + (buf) * BM * KB + (wu * A_LI + it) * 1024), 16, 0, 0);  \
            _Pragma("unroll") for (int it = 0; it < W_LI; ++it)                                                            \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[it] + (k0)),        \
                    (__attribute__((address_space(3))) void*)(Ws + (buf) * BN * KB + (wu * W_LI + it) * 1024), 16, 0, 0);  \
        }
        const int g4 = lane >> 4, sx = frow & 7;
        const int a_off = (wm * (BM / WM) + frow) * KB, w_off = (wn * (BN / WN) + frow) * KB;
        {
        PQ_DLOAD(0, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) PQ_DLOAD(cur ^ 1, (kt + 1) * BK)
            const unsigned char* Ab = As + cur * BM * KB + a_off;
            const unsigned char* Wb = Ws + cur * BN * KB + w_off;
            const unsigned char* Pb = tr ? Ab : Wb;
            const unsigned char* Qb = tr ? Wb : Ab;
            if constexpr (PAIRS) {
                const int so_h = (g4 ^ sx) * 16, so_l = ((4 + g4) ^ sx) * 16;
                Frag<T> ph[TM], pl[TM], qh[TN], ql[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ph[i].v = *reinterpret_cast<const decltype(ph[i].v)*>(Pb + i * 16 * KB + so_h);
                    pl[i].v = *reinterpret_cast<const decltype(pl[i].v)*>(Pb + i * 16 * KB + so_l);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    qh[j].v = *reinterpret_cast<const decltype(qh[j].v)*>(Qb + j * 16 * KB + so_h);
                    ql[j].v = *reinterpret_cast<const decltype(ql[j].v)*>(Qb + j * 16 * KB + so_l);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        mma16(acc[i][j], pl[i], qh[j]);          // small terms first
                        mma16(acc[i][j], ph[i], ql[j]);
                        mma16(acc[i][j], ph[i], qh[j]);
                    }
            } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int so = ((kk * 4 + g4) ^ sx) * 16;
                Frag<T> fp[TM], fq[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fp[i].v = *reinterpret_cast<const decltype(fp[i].v)*>(Pb + i * 16 * KB + so);
#pragma unroll
                for (int j = 0; j < TN; ++j) fq[j].v = *reinterpret_cast<const decltype(fq[j].v)*>(Qb + j * 16 * KB + so);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mma16(acc[i][j], fp[i], fq[j]);
            }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA into the other buffer has landed
            __syncthreads();                                     // ... and so has everybody else's; buffer `cur` is free
        }
        }
#undef PQ_DLOAD
    } else {
    // Staging: each thread owns A_IT + W_IT 16-byte chunks of a stage.  Loads are unconditional from clamped addresses
    // (no divergent control flow, everything stays in registers); out-of-range chunks are zeroed by a select.
    typename ALoad::Raw ra[A_IT];
    u32x4 rw[W_IT];
    int a_row[A_IT], w_row[W_IT], a_col[A_IT], w_col[W_IT];
    bool a_ok[A_IT], w_ok[W_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int c = it * NT + tid, r = m0 + c / CPR;
        a_ok[it] = r < M; a_row[it] = a_ok[it] ? r : M - 1; a_col[it] = (c % CPR) * EPC;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int c = it * NT + tid, r = n0 + c / CPR;
        w_ok[it] = r < N; w_row[it] = w_ok[it] ? r : N - 1; w_col[it] = (c % CPR) * EPC;
    }
    const int klast = K - EPC;
    const u32x4 zero = {0u, 0u, 0u, 0u};

    // PQ_GLOAD only issues loads (clamped addresses, no use of the data); PQ_LSTORE — placed AFTER the stage's MFMAs —
    // converts / zero-fills and writes LDS.  Keeping every consumer of the loaded registers behind the MFMAs is what lets
    // the load latency overlap the matrix work (a select right after the load would force an s_waitcnt in front of them).
#define PQ_GLOAD(k0)                                                                                  \
    {                                                                                                 \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                         \
            const int k_ = (k0) + a_col[it];                                                          \
            ra[it] = aload.fetch(a_row[it], k_ < K ? k_ : klast);                                     \
        }                                                                                             \
        _Pragma("unroll") for (int it = 0; it < W_IT; ++it) {                                         \
            const int k_ = (k0) + w_col[it];                                                          \
            rw[it] = *reinterpret_cast<const u32x4*>(W + (size_t)w_row[it] * ldw + (k_ < K ? k_ : klast)); \
        }                                                                                             \
    }
#define PQ_LSTORE(buf, k0)                                                                                             \
    {                                                                                                                  \
        _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                                          \
            const int c_ = it * NT + tid, k_ = (k0) + a_col[it];                                                       \
            const u32x4 v0_ = aload.finish(ra[it], a_row[it], k_ < K ? k_ : klast);                                    \
            const u32x4 v_ = (a_ok[it] && k_ < K) ? v0_ : zero;                                                        \
            if constexpr (SPLIT) {      /* chunk q of the stage row: block q / 8, hi at 8 (q % 8), lo 64 bytes further */ \
                uint2 hi_, lo_;                                                                                        \
                split4(v_, hi_, lo_);                                                                                  \
                unsigned char* d_ = As + ((buf) * BM + c_ / CPR) * GEMM_ROWB + ((c_ % CPR) >> 3) * 128 + ((c_ % CPR) & 7) * 8; \
                *reinterpret_cast<uint2*>(d_) = hi_;                                                                   \
                *reinterpret_cast<uint2*>(d_ + 64) = lo_;                                                              \
            } else {                                                                                                   \
                *reinterpret_cast<u32x4*>(As + ((buf) * BM + c_ / CPR) * GEMM_ROWB + (c_ % CPR) * 16) = v_;             \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int it = 0; it < W_IT; ++it) {                                                          \
            const int c_ = it * NT + tid, k_ = (k0) + w_col[it];                                                       \
            *reinterpret_cast<u32x4*>(Ws + ((buf) * BN + c_ / CPR) * GEMM_ROWB + (c_ % CPR) * 16) =                     \
                (w_ok[it] && k_ < K) ? rw[it] : zero;                                                                  \
        }                                                                                                              \
    }

    PQ_GLOAD(0)
    PQ_LSTORE(0, 0)
    __syncthreads();

    const int fk = (lane >> 4) * 16;
    const int a_off = (wm * (BM / WM) + frow) * GEMM_ROWB + fk;
    const int w_off = (wn * (BN / WN) + frow) * GEMM_ROWB + fk;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (NBUF == 2) ? (kt & 1) : 0;
        if (kt + 1 < nk) PQ_GLOAD((kt + 1) * BK)
        const unsigned char* Ab = As + cur * BM * GEMM_ROWB + a_off;
        const unsigned char* Wb = Ws + cur * BN * GEMM_ROWB + w_off;
        const unsigned char* Pb = tr ? Ab : Wb;
        const unsigned char* Qb = tr ? Wb : Ab;
        if constexpr (SPLIT) {
#pragma unroll
            for (int kb = 0; kb < KB / 128; ++kb) {          // one 32-element block: hi plane at +0, lo plane at +64
                Frag<bf16_t> ph[TM], pl[TM], qh[TN], ql[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ph[i].v = *reinterpret_cast<const bf16x8*>(Pb + i * 16 * GEMM_ROWB + kb * 128);
                    pl[i].v = *reinterpret_cast<const bf16x8*>(Pb + i * 16 * GEMM_ROWB + kb * 128 + 64);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    qh[j].v = *reinterpret_cast<const bf16x8*>(Qb + j * 16 * GEMM_ROWB + kb * 128);
                    ql[j].v = *reinterpret_cast<const bf16x8*>(Qb + j * 16 * GEMM_ROWB + kb * 128 + 64);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        mma16(acc[i][j], pl[i], qh[j]);          // small terms first
                        mma16(acc[i][j], ph[i], ql[j]);
                        mma16(acc[i][j], ph[i], qh[j]);
                    }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < KB / 64; ++kk) {
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
        }
        if (kt + 1 < nk) {
            if constexpr (NBUF == 1) __syncthreads();        // every wave has finished reading the single stage
            PQ_LSTORE((NBUF == 2) ? (cur ^ 1) : 0, (kt + 1) * BK)
        }
        __syncthreads();
    }

    }
#undef PQ_GLOAD
#undef PQ_LSTORE
    // ---- epilogue: registers -> (phase 1 math) -> LDS tile in the output's storage type -> coalesced 16-byte stores.
    // The main loop ended on a barrier, so the stage buffers are free to be reused as the staging tile.
    using S = typename Epi::S;
    constexpr int CH = 16 / (int)sizeof(S);
    unsigned char* St = smem;
    const int wmb = wm * (BM / WM), wnb = wn * (BN / WN);
    if (!tr) {
        constexpr int SROW = BN * (int)sizeof(S) + 16;
        constexpr int CPRO = BN / CH;
        constexpr int NCH = (BM * CPRO + NT - 1) / NT;              // chunks per thread
        u32x4 pre[Epi::kPrefetch ? NCH : 1];
        if constexpr (Epi::kPrefetch) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = q * NT + tid, r = c / CPRO, cc = c - r * CPRO;
                const int mm = m0 + r < M ? m0 + r : M - 1, nn = n0 + cc * CH + CH <= N ? n0 + cc * CH : 0;
                pre[q] = epi.load_n(mm, nn);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                const int ml = wmb + j * 16 + (lane & 15), nl = wnb + i * 16 + 4 * (lane >> 4);
                epi.xform_n4(m0 + ml, n0 + nl, v);
                store4<S>(reinterpret_cast<S*>(St + ml * SROW) + nl, v);
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = q * NT + tid, r = c / CPRO, cc = c - r * CPRO;
            if (c < BM * CPRO && m0 + r < M && n0 + cc * CH < N) {
                const S* chunk = reinterpret_cast<const S*>(St + r * SROW + cc * 16);
                if constexpr (Epi::kPrefetch) epi.store_n(m0 + r, n0 + cc * CH, chunk, pre[q]);
                else epi.store_n(m0 + r, n0 + cc * CH, chunk);
            }
        }
    } else {
        constexpr int SROW = BM * (int)sizeof(S) + 16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                const int ml = wmb + i * 16 + 4 * (lane >> 4), nl = wnb + j * 16 + (lane & 15);
                epi.xform_m4(m0 + ml, n0 + nl, v);
                store4<S>(reinterpret_cast<S*>(St + nl * SROW) + ml, v);
            }
        __syncthreads();
        constexpr int CPRO = BM / CH;
        for (int c = tid; c < BN * CPRO; c += NT) {
            const int r = c / CPRO, cc = c - r * CPRO;
            if (n0 + r < N && m0 + cc * CH < M)
                epi.store_m(m0 + cc * CH, n0 + r, reinterpret_cast<const S*>(St + r * SROW + cc * 16));
        }
    }
}

template <int BM, int BN, int KB, int NBUF, int STATS, int SSZ>
constexpr size_t gemm_lds_bytes() {
    const size_t pipe = (size_t)NBUF * (BM + BN) * gemm_rowb<KB>() + (size_t)BM * STATS * sizeof(float);
    const size_t stage_n = (size_t)BM * (BN * SSZ + 16), stage_m = (size_t)BN * (BM * SSZ + 16);   // epilogue staging tile
    const size_t stage = stage_n > stage_m ? stage_n : stage_m;
    return pipe > stage ? pipe : stage;
}

// Both operands pre-split (gemm_kernel PAIRS): A and W are bf16 views [M][2 K_logical] / [N][2 K_logical]; K = 2 K_logical must be a
// multiple of 64 (whole 32-element blocks of the logical matrices).
template <int BM, int BN, int WM, int WN, typename Epi>
inline hipError_t launch_gemm_pairs(hipStream_t s, const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const Epi& epi) {
    const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
    const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
    constexpr size_t lds = gemm_lds_bytes<BM, BN, 128, 2, 0, (int)sizeof(typename Epi::S)>();
    if (K % 64) return hipErrorInvalidValue;
    auto kd = gemm_kernel<bf16_t, BM, BN, WM, WN, 128, 2, true, ARowMajor<bf16_t>, Epi, false, true>;
    if (lds > 64 * 1024) {
        static LdsAttr attr_d;
        if (hipError_t e = attr_d.ensure(reinterpret_cast<const void*>(kd), lds); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kd, dim3(grid), dim3(WM * WN * 64), lds, s, ARowMajor<bf16_t>{A, lda}, W, ldw, M, N, K, mtiles, ntiles, epi);
    return hipGetLastError();
}

template <typename T, int BM, int BN, int WM, int WN, int KB, int NBUF, typename ALoad, typename Epi, bool SPLIT = false>
inline hipError_t launch_gemm(hipStream_t s, const ALoad& aload, const T* W, int ldw, int M, int N, int K, const Epi& epi) {
    const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
    const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
    constexpr size_t lds = gemm_lds_bytes<BM, BN, KB, NBUF, ALoad::kStatsFloats, (int)sizeof(typename Epi::S)>();
    constexpr bool can_direct = !SPLIT && ALoad::kDirect && KB == 128 && NBUF == 2 && (BM % (8 * WM * WN) == 0) && (BN % (8 * WM * WN) == 0);
    if constexpr (can_direct) {
        if (K % (KB / (int)sizeof(T)) == 0) {       // whole stages only: the DMA path cannot zero-fill a K tail
            auto kd = gemm_kernel<T, BM, BN, WM, WN, KB, NBUF, true, ALoad, Epi>;
            if (lds > 64 * 1024) {
                static LdsAttr attr_d;
                if (hipError_t e = attr_d.ensure(reinterpret_cast<const void*>(kd), lds); e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(kd, dim3(grid), dim3(WM * WN * 64), lds, s, aload, W, ldw, M, N, K, mtiles, ntiles, epi);
            return hipGetLastError();
        }
    }
    auto kern = gemm_kernel<T, BM, BN, WM, WN, KB, NBUF, false, ALoad, Epi, SPLIT>;
    // diagnostic (tools/x3_diag2.py): PARSEQ_GEMM_EXTRA_LDS=<bytes> pads the dynamic LDS request, e.g. to keep a second workgroup off the CU
    static const size_t extra = getenv("PARSEQ_GEMM_EXTRA_LDS") ? (size_t)atol(getenv("PARSEQ_GEMM_EXTRA_LDS")) : 0;
    const size_t lds_req = lds + extra;
    if (lds_req > 64 * 1024) {
        static LdsAttr attr;                // one per template instantiation
        if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds_req); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds_req, s, aload, W, ldw, M, N, K, mtiles, ntiles, epi);
    return hipGetLastError();
}

}  // namespace pq
