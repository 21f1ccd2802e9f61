// The training step's all-bf16 products (train_ops.h mfma_bgemm16_kernel: both operands bf16 and k-contiguous in memory) timed ALONE, without
// torch, on the encoder's shapes — to separate a tile's fixed cost (prologue + epilogue) from its per-stage cost and the stores from the rest:
//   * K sweep at fixed M x N (K = 64 is ONE 64-deep stage): T(K) = fixed + per-stage * K / 64;
//   * epilogue forms: no store at all (C = c16 = nullptr), bf16 only, fp32 only, fp32 + bf16, bf16 pre-activation + bf16 GELU (fc1's form);
//   * the three-workgroup (WHOLE = false) and four-workgroup (WHOLE = true) forms of the kernel;
//   * hot (the same buffers every launch: operands and outputs live in the 256 MB Infinity Cache) vs cold (four copies in rotation);
//   * the dW form (mfma_bgemm16t_kernel) over its split counts.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iparseq_amd/csrc -o tools/microbench/train_gemm tools/microbench/train_gemm.hip && tools/microbench/train_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"
#include "rowops.h"
#include "train_ops.h"
using namespace pq;

// mfma_bgemm16_kernel<true> again with pieces cut out (MODE): 0 whole, 1 no epilogue, 2 no main loop (epilogue of zeros), 3 empty kernel,
// 4 main loop without the MFMAs (loads, LDS stores, barriers), 5 main loop without the global loads (MFMAs on whatever LDS holds)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ablate_kernel(const SgemmArgs a, int k_chunk, float* __restrict__ partial, int gn, int gm) {
    constexpr int TILE_BYTES = MG_BM * BH_LD * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    if constexpr (MODE == 3) { if (a.M < 0) partial[threadIdx.x] = 1.f; return; }
    bf16_t (*As)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem);
    bf16_t (*Bs)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem + TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BgTile bt = bg_tile<false>(gn, gm);
    const int m0 = bt.tm * MG_BM, n0 = bt.tn * MG_BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kbeg = 0, kend = a.K;
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, 0x7FFFF000, 0x00020000);
    const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, 0x7FFFF000, 0x00020000);
    const unsigned oa = 2u * ((unsigned)(m0 + (tid >> 3)) * (unsigned)a.sam + 8u * (tid & 7));
    const unsigned ob = 2u * ((unsigned)(n0 + (tid >> 3)) * (unsigned)a.sbn + 8u * (tid & 7));
    const unsigned pa_step = 64u * (unsigned)a.sam, pb_step = 64u * (unsigned)a.sbn;
    unsigned kbyte = 0;
    u32x4 ra[4] = {}, rb[4] = {};
    auto fetch = [&]() {
        if constexpr (MODE != 5) {
#pragma unroll
            for (int it = 0; it < 4; ++it) ra[it] = __builtin_amdgcn_raw_buffer_load_b128(ares, oa, kbyte + it * pa_step, 0);
#pragma unroll
            for (int it = 0; it < 4; ++it) rb[it] = __builtin_amdgcn_raw_buffer_load_b128(bres, ob, kbyte + it * pb_step, 0);
        }
        kbyte += 2u * BH_BK;
    };
    if constexpr (MODE != 2) {
        fetch();
        for (int k0 = kbeg; k0 < kend; k0 += BH_BK) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + 256 * it;
                *reinterpret_cast<u32x4*>(&As[idx >> 3][8 * (idx & 7)]) = ra[it];
                *reinterpret_cast<u32x4*>(&Bs[idx >> 3][8 * (idx & 7)]) = rb[it];
            }
            __syncthreads();
            if (k0 + BH_BK < kend) fetch();
            if constexpr (MODE != 4) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) bg16_stage_mfma<true>(As, Bs, acc, wm, wn, r16, g, kk);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
    if constexpr (MODE == 1 || MODE == 4 || MODE == 5) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (MODE == 4) t += __uint_as_float(ra[0][0] ^ rb[3][3]) + reinterpret_cast<const float*>(smem)[tid];
        if (t == 1.2345e-33f) partial[tid] = t;      // never true: keeps the work alive
    } else {
        bg_epilogue(a, acc, partial, reinterpret_cast<float*>(smem), m0, n0, tid, 0);
    }
}

// Candidate for round 4 (NOT in the library): the same product computed TRANSPOSED — MFMA operands swapped, so a lane holds D[m = 16 i + r16][n = 16 j + 4 g
// + r], four consecutive columns of ONE row — and stored straight from the accumulators as 16-byte (fp32) / 8-byte (bf16) pieces: no trip through LDS.
// Forms: bias, fp32 C and / or bf16 c16 and / or bf16 GELU; no residual, no accumulate, no split.  main() compares its outputs with the library kernel's bit for bit.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void cand_transposed_kernel(const SgemmArgs a, int k_chunk, float* __restrict__ partial, int gn, int gm) {
    constexpr int TILE_BYTES = MG_BM * BH_LD * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    bf16_t (*As)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem);
    bf16_t (*Bs)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem + TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BgTile bt = bg_tile<false>(gn, gm);
    const int m0 = bt.tm * MG_BM, n0 = bt.tn * MG_BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, 0x7FFFF000, 0x00020000);
    const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, 0x7FFFF000, 0x00020000);
    const unsigned oa = 2u * ((unsigned)(m0 + (tid >> 3)) * (unsigned)a.sam + 8u * (tid & 7));
    const unsigned ob = 2u * ((unsigned)(n0 + (tid >> 3)) * (unsigned)a.sbn + 8u * (tid & 7));
    const unsigned pa_step = 64u * (unsigned)a.sam, pb_step = 64u * (unsigned)a.sbn;
    unsigned kbyte = 0;
    u32x4 ra[4], rb[4];
    auto fetch = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) ra[it] = __builtin_amdgcn_raw_buffer_load_b128(ares, oa, kbyte + it * pa_step, 0);
#pragma unroll
        for (int it = 0; it < 4; ++it) rb[it] = __builtin_amdgcn_raw_buffer_load_b128(bres, ob, kbyte + it * pb_step, 0);
        kbyte += 2u * BH_BK;
    };
    fetch();
    for (int k0 = 0; k0 < a.K; k0 += BH_BK) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 256 * it;
            *reinterpret_cast<u32x4*>(&As[idx >> 3][8 * (idx & 7)]) = ra[it];
            *reinterpret_cast<u32x4*>(&Bs[idx >> 3][8 * (idx & 7)]) = rb[it];
        }
        __syncthreads();
        if (k0 + BH_BK < a.K) fetch();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const bf16x8*>(&As[wm + 16 * i + r16][32 * kk + 8 * g]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bj = *reinterpret_cast<const bf16x8*>(&Bs[wn + 16 * j + r16][32 * kk + 8 * g]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bj, av[i], acc[i][j], 0, 0, 0);      // D^T: rows = n, columns = m
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    // lane: row m = m0 + wm + 16 i + r16, columns n0 + wn + 16 j + 4 g .. + 3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gn0 = n0 + wn + 16 * j + 4 * g;
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + gn0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t at = (size_t)(m0 + wm + 16 * i + r16) * a.ldc + gn0;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(a.alpha, acc[i][j][e], b4[e]);
            if (a.C) *reinterpret_cast<f32x4*>(a.C + at) = o;
            if (a.c16) {
                union { u32x2 u; bf16_t e[4]; } hh;
#pragma unroll
                for (int e = 0; e < 4; ++e) hh.e[e] = static_cast<bf16_t>(o[e]);
                *reinterpret_cast<u32x2*>(a.c16 + at) = hh.u;
            }
            if (a.gelu_out16) {
                union { u32x2 u; bf16_t e[4]; } hh;
#pragma unroll
                for (int e = 0; e < 4; ++e) hh.e[e] = static_cast<bf16_t>(gelu_erf(o[e]));
                *reinterpret_cast<u32x2*>(a.gelu_out16 + at) = hh.u;
            }
        }
    }
}
// Candidate 2 (NOT in the library; measured with the round's last GPU seconds: bit-identical, no faster — profiles/r03_train_gemm_microbench.md): the transposed product again, but the
// accumulators still go through LDS as in bg_epilogue — only now a lane holds four consecutive COLUMNS of a row, so the staging pass is 16
// `ds_write_b128` per lane and half instead of 64 `ds_write_b32` (the epilogue alone, nothing stored, was 27.5 us of the 75 us 49 152 x 1536 x 384
// product).  Read-back and stores are bg_epilogue's vec path for the forms below (bias, fp32 C and / or bf16 c16 and / or bf16 GELU).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void cand_staged_transposed_kernel(const SgemmArgs a, int k_chunk, float* __restrict__ partial, int gn, int gm) {
    constexpr int TILE_BYTES = MG_BM * BH_LD * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    bf16_t (*As)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem);
    bf16_t (*Bs)[BH_LD] = reinterpret_cast<bf16_t (*)[BH_LD]>(smem + TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BgTile bt = bg_tile<false>(gn, gm);
    const int m0 = bt.tm * MG_BM, n0 = bt.tn * MG_BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, 0x7FFFF000, 0x00020000);
    const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, 0x7FFFF000, 0x00020000);
    const unsigned oa = 2u * ((unsigned)(m0 + (tid >> 3)) * (unsigned)a.sam + 8u * (tid & 7));
    const unsigned ob = 2u * ((unsigned)(n0 + (tid >> 3)) * (unsigned)a.sbn + 8u * (tid & 7));
    const unsigned pa_step = 64u * (unsigned)a.sam, pb_step = 64u * (unsigned)a.sbn;
    unsigned kbyte = 0;
    u32x4 ra[4], rb[4];
    auto fetch = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) ra[it] = __builtin_amdgcn_raw_buffer_load_b128(ares, oa, kbyte + it * pa_step, 0);
#pragma unroll
        for (int it = 0; it < 4; ++it) rb[it] = __builtin_amdgcn_raw_buffer_load_b128(bres, ob, kbyte + it * pb_step, 0);
        kbyte += 2u * BH_BK;
    };
    fetch();
    for (int k0 = 0; k0 < a.K; k0 += BH_BK) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 256 * it;
            *reinterpret_cast<u32x4*>(&As[idx >> 3][8 * (idx & 7)]) = ra[it];
            *reinterpret_cast<u32x4*>(&Bs[idx >> 3][8 * (idx & 7)]) = rb[it];
        }
        __syncthreads();
        if (k0 + BH_BK < a.K) fetch();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const bf16x8*>(&As[wm + 16 * i + r16][32 * kk + 8 * g]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bj = *reinterpret_cast<const bf16x8*>(&Bs[wn + 16 * j + r16][32 * kk + 8 * g]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bj, av[i], acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    float* stage = reinterpret_cast<float*>(smem);
    const int c4 = tid & 31, gcol = n0 + 4 * c4;
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + gcol);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((wave >> 1) == h) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(stage + (16 * i + r16) * EP_LD + wn + 16 * j + 4 * g) = acc[i][j];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = (tid >> 5) + 8 * q;
            const size_t at = (size_t)(m0 + 64 * h + row) * a.ldc + gcol;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * EP_LD + 4 * c4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(a.alpha, v[e], b4[e]);
            if (a.C) *reinterpret_cast<f32x4*>(a.C + at) = o;
            if (a.c16) {
                union { u32x2 u; bf16_t e[4]; } hh;
#pragma unroll
                for (int e = 0; e < 4; ++e) hh.e[e] = static_cast<bf16_t>(o[e]);
                *reinterpret_cast<u32x2*>(a.c16 + at) = hh.u;
            }
            if (a.gelu_out16) {
                union { u32x2 u; bf16_t e[4]; } hh;
#pragma unroll
                for (int e = 0; e < 4; ++e) hh.e[e] = static_cast<bf16_t>(gelu_erf(o[e]));
                *reinterpret_cast<u32x2*>(a.gelu_out16 + at) = hh.u;
            }
        }
        __syncthreads();
    }
}
__global__ void count_diff_kernel(const unsigned* x, const unsigned* y, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned d = 0;
    for (; i < n; i += (size_t)gridDim.x * 256) d += x[i] != y[i];
    if (d) atomicAdd(out, d);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ROT > 1: every launch takes the next of ROT copies of the A operand and of the outputs, so that neither comes out of the 256 MB Infinity Cache
// (what the products see inside the step); grid z = splits of the contraction (the dW products; partials go to `scratch`)
struct Bufs { bf16_t* A[4]; bf16_t* c16[4]; bf16_t* g16[4]; float* C[4]; };
static float run(void (*kern)(const SgemmArgs, int, float*, int, int), SgemmArgs a, const Bufs& b, bool c32, bool c16, bool g16, int rot, float* scratch,
                 int iters, int splits = 1, bool a_is_rot = true) {
    const int gm = (a.M + MG_BM - 1) / MG_BM, gn = (a.N + MG_BN - 1) / MG_BN;
    const int k_chunk = ((a.K + splits - 1) / splits + 63) / 64 * 64;
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto go = [&](int i) {
        const int r = i % rot;
        if (a_is_rot) a.A = reinterpret_cast<const float*>(b.A[r]); else a.B = reinterpret_cast<const float*>(b.A[r]);
        a.C = c32 ? b.C[r] : nullptr; a.c16 = c16 ? b.c16[r] : nullptr; a.gelu_out16 = g16 ? b.g16[r] : nullptr;
        hipLaunchKernelGGL(kern, dim3(gm * gn, 1, splits), dim3(256), 0, 0, a, k_chunk, scratch, gn, gm);
    };
    for (int i = 0; i < 4; ++i) go(i);
    CK(hipEventRecord(t0, 0));
    for (int i = 0; i < iters; ++i) go(i);
    CK(hipEventRecord(t1, 0));
    CK(hipEventSynchronize(t1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, t0, t1));
    return 1e3f * ms / iters;      // us per launch
}

int main(int argc, char** argv) {
    const int M = 49152, iters = 20, ROT = 4;
    const int KMAX = 1536, NMAX = 1536;
    std::vector<unsigned short> h((size_t)M * KMAX);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    Bufs b; bf16_t* B; float *bias, *scratch;
    for (int r = 0; r < ROT; ++r) {
        CK(hipMalloc(&b.A[r], (size_t)M * KMAX * 2)); CK(hipMalloc(&b.c16[r], (size_t)M * NMAX * 2)); CK(hipMalloc(&b.g16[r], (size_t)M * NMAX * 2));
        CK(hipMalloc(&b.C[r], (size_t)M * NMAX * 4));
        CK(hipMemcpy(b.A[r], h.data(), (size_t)M * KMAX * 2, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&B, (size_t)M * 384 * 2)); CK(hipMalloc(&bias, NMAX * 4)); CK(hipMalloc(&scratch, (size_t)64 << 20));
    CK(hipMemcpy(B, h.data(), (size_t)M * 384 * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, NMAX * 4));
    struct Epi { const char* name; bool c32, c16, g16; };
    const Epi epis[] = {{"no store", false, false, false}, {"bf16", false, true, false}, {"fp32", true, false, false}, {"fp32+bf16", true, true, false},
                        {"bf16+gelu16", false, true, true}};
    printf("forward / dX form (mfma_bgemm16_kernel): hot = the same buffers every launch, cold = %d copies of A and of the outputs in rotation\n\n", ROT);
    printf("| M x N x K | epilogue | 3 wg/CU hot us | 4 wg/CU hot us | 4 wg/CU cold us | TFLOP/s (4, cold) | tiles |\n|---|---|---:|---:|---:|---:|---:|\n");
    const int shapes[][2] = {{384, 64}, {384, 384}, {384, 1536}, {1152, 384}, {1536, 64}, {1536, 384}};
    for (const auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        for (const Epi& e : epis) {
            if (K == 64 && (e.c32 && e.c16)) continue;
            SgemmArgs a{};
            a.sam = K; a.sak = 1; a.B = reinterpret_cast<const float*>(B); a.sbk = 1; a.sbn = K;
            a.bias = bias; a.R = nullptr; a.ldr = 0; a.rper = 1; a.ldc = N; a.M = M; a.N = N; a.K = K; a.alpha = 1.f; a.a16 = a.b16 = 1;
            const float t3 = run(mfma_bgemm16_kernel<false>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters);
            const float t4 = run(mfma_bgemm16_kernel<true>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters);
            const float t4c = run(mfma_bgemm16_kernel<true>, a, b, e.c32, e.c16, e.g16, ROT, scratch, iters);
            printf("| %d x %d x %d | %s | %.1f | %.1f | %.1f | %.0f | %d |\n", M, N, K, e.name, t3, t4, t4c, 2.0 * M * N * K / t4c * 1e-6, (M / 128) * (N / 128));
        }
    }
    printf("\nwhere a tile's time goes (mfma_bgemm16_kernel<true> with pieces cut out, hot, four workgroups per CU)\n\n");
    printf("| M x N x K | epilogue | whole | no epilogue | no main loop | loop without MFMAs | loop without loads | empty kernel |\n|---|---|---:|---:|---:|---:|---:|---:|\n");
    const int ab[][2] = {{384, 64}, {384, 384}, {1536, 384}, {384, 1536}};
    for (const auto& sh : ab) {
        const int N = sh[0], K = sh[1];
        for (int ei = 0; ei < 2; ++ei) {
            const Epi& e = epis[ei];
            SgemmArgs a{};
            a.sam = K; a.sak = 1; a.B = reinterpret_cast<const float*>(B); a.sbk = 1; a.sbn = K;
            a.bias = bias; a.R = nullptr; a.ldr = 0; a.rper = 1; a.ldc = N; a.M = M; a.N = N; a.K = K; a.alpha = 1.f; a.a16 = a.b16 = 1;
            const float t0 = run(ablate_kernel<0>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters), t1 = run(ablate_kernel<1>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters);
            const float t2 = run(ablate_kernel<2>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters), t4 = run(ablate_kernel<4>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters);
            const float t5 = run(ablate_kernel<5>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters), t3 = run(ablate_kernel<3>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters);
            printf("| %d x %d x %d | %s | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |\n", M, N, K, e.name, t0, t1, t2, t4, t5, t3);
        }
    }
    printf("\ncandidate: transposed product, stores straight from the accumulators (vs mfma_bgemm16_kernel<true>; differing 32-bit words of the outputs)\n\n");
    printf("candidate 2: the transposed product with bg_epilogue's LDS staging, written as 16-byte pieces\n\n");
    printf("| M x N x K | epilogue | library hot | candidate hot | library cold | candidate cold | words that differ | cand. 2 hot | cand. 2 cold | cand. 2 words that differ |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|\n");
    unsigned* dcount; CK(hipMalloc(&dcount, 4));
    const int cs[][2] = {{384, 384}, {1152, 384}, {1536, 384}, {384, 1536}};
    for (const auto& sh : cs) {
        const int N = sh[0], K = sh[1];
        for (int ei = 1; ei < 5; ++ei) {
            const Epi& e = epis[ei];
            SgemmArgs a{};
            a.sam = K; a.sak = 1; a.B = reinterpret_cast<const float*>(B); a.sbk = 1; a.sbn = K;
            a.bias = bias; a.R = nullptr; a.ldr = 0; a.rper = 1; a.ldc = N; a.M = M; a.N = N; a.K = K; a.alpha = 1.f; a.a16 = a.b16 = 1;
            // correctness first: library -> copy 0 of the outputs, candidate -> copy 1, same operands
            Bufs b0 = b, b1 = b;
            b1.C[0] = b.C[1]; b1.c16[0] = b.c16[1]; b1.g16[0] = b.g16[1]; b1.A[0] = b.A[0];
            CK(hipMemset(b.C[0], 0xFF, (size_t)M * N * 4)); CK(hipMemset(b.C[1], 0xEE, (size_t)M * N * 4));
            CK(hipMemset(b.c16[0], 0xFF, (size_t)M * N * 2)); CK(hipMemset(b.c16[1], 0xEE, (size_t)M * N * 2));
            CK(hipMemset(b.g16[0], 0xFF, (size_t)M * N * 2)); CK(hipMemset(b.g16[1], 0xEE, (size_t)M * N * 2));
            run(mfma_bgemm16_kernel<true>, a, b0, e.c32, e.c16, e.g16, 1, scratch, 1);
            run(cand_transposed_kernel, a, b1, e.c32, e.c16, e.g16, 1, scratch, 1);
            CK(hipMemset(dcount, 0, 4));
            if (e.c32) hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)b.C[0], (const unsigned*)b.C[1], (size_t)M * N, dcount);
            if (e.c16) hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)b.c16[0], (const unsigned*)b.c16[1], (size_t)M * N / 2, dcount);
            if (e.g16) hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)b.g16[0], (const unsigned*)b.g16[1], (size_t)M * N / 2, dcount);
            unsigned diff = 0; CK(hipMemcpy(&diff, dcount, 4, hipMemcpyDeviceToHost));
            const float l1 = run(mfma_bgemm16_kernel<true>, a, b, e.c32, e.c16, e.g16, 1, scratch, iters), c1 = run(cand_transposed_kernel, a, b, e.c32, e.c16, e.g16, 1, scratch, iters);
            const float l4 = run(mfma_bgemm16_kernel<true>, a, b, e.c32, e.c16, e.g16, ROT, scratch, iters), c4 = run(cand_transposed_kernel, a, b, e.c32, e.c16, e.g16, ROT, scratch, iters);
            // candidate 2: outputs into copy 2, compared with the library's copy 0
            Bufs b2 = b; b2.C[0] = b.C[2]; b2.c16[0] = b.c16[2]; b2.g16[0] = b.g16[2];
            CK(hipMemset(b.C[2], 0xDD, (size_t)M * N * 4)); CK(hipMemset(b.c16[2], 0xDD, (size_t)M * N * 2)); CK(hipMemset(b.g16[2], 0xDD, (size_t)M * N * 2));
            run(mfma_bgemm16_kernel<true>, a, b0, e.c32, e.c16, e.g16, 1, scratch, 1);
            run(cand_staged_transposed_kernel, a, b2, e.c32, e.c16, e.g16, 1, scratch, 1);
            CK(hipMemset(dcount, 0, 4));
            if (e.c32) hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)b.C[0], (const unsigned*)b.C[2], (size_t)M * N, dcount);
            if (e.c16) hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)b.c16[0], (const unsigned*)b.c16[2], (size_t)M * N / 2, dcount);
            if (e.g16) hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)b.g16[0], (const unsigned*)b.g16[2], (size_t)M * N / 2, dcount);
            unsigned diff2 = 0; CK(hipMemcpy(&diff2, dcount, 4, hipMemcpyDeviceToHost));
            const float s1 = run(cand_staged_transposed_kernel, a, b, e.c32, e.c16, e.g16, 1, scratch, iters), s4 = run(cand_staged_transposed_kernel, a, b, e.c32, e.c16, e.g16, ROT, scratch, iters);
            printf("| %d x %d x %d | %s | %.1f | %.1f | %.1f | %.1f | %u | %.1f | %.1f | %u |\n", M, N, K, e.name, l1, c1, l4, c4, diff, s1, s4, diff2);
        }
    }
    // dW form: C[Nout, Kin] = dY^T X, both operands outer-contiguous bf16 (dY [rows, Nout], X [rows, Kin]), contraction over the 49 152 rows in `splits`
    // workgroups along z writing fp32 partials (the fold is a separate kernel, not timed here)
    printf("\ndW form (mfma_bgemm16t_kernel, partials only): dY [49152, Nout] (rotated when cold), X [49152, 384]\n\n");
    printf("| Nout x Kin | splits | workgroups | 3 wg/CU hot us | 4 wg/CU hot us | 4 wg/CU cold us | TFLOP/s (4, cold) |\n|---|---:|---:|---:|---:|---:|---:|\n");
    const int dws[][2] = {{1536, 15}, {1536, 28}, {1152, 19}, {384, 57}, {384, 110}};
    for (const auto& d : dws) {
        const int Nout = d[0], splits = d[1], Kin = 384;
        SgemmArgs a{};
        a.sam = 1; a.sak = Nout; a.B = reinterpret_cast<const float*>(B); a.sbk = Kin; a.sbn = 1;
        a.ldc = Kin; a.M = Nout; a.N = Kin; a.K = M; a.alpha = 1.f; a.a16 = a.b16 = 1; a.rper = 1;
        const float t3 = run(mfma_bgemm16t_kernel<false>, a, b, true, false, false, 1, scratch, iters, splits);
        const float t4 = run(mfma_bgemm16t_kernel<true>, a, b, true, false, false, 1, scratch, iters, splits);
        const float t4c = run(mfma_bgemm16t_kernel<true>, a, b, true, false, false, ROT, scratch, iters, splits);
        printf("| %d x %d | %d | %d | %.1f | %.1f | %.1f | %.0f |\n", Nout, Kin, splits, (Nout / 128) * (Kin / 128) * splits, t3, t4, t4c, 2.0 * M * Nout * Kin / t4c * 1e-6);
    }
    return 0;
}
