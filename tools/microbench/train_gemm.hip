// The training step's all-bf16 products (train_ops.h mfma_bgemm16_kernel: both operands bf16 and k-contiguous in memory) timed ALONE, without
// torch, on the encoder's shapes — to separate a tile's fixed cost (prologue + epilogue) from its per-stage cost and the stores from the rest:
//   * K sweep at fixed M x N (K = 64 is ONE 64-deep stage): T(K) = fixed + per-stage * K / 64;
//   * epilogue forms: no store at all (C = c16 = nullptr), bf16 only, fp32 only, fp32 + bf16, bf16 pre-activation + bf16 GELU (fc1's form);
//   * the three-workgroup (WHOLE = false) and four-workgroup (WHOLE = true) forms of the kernel.
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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float run(void (*kern)(const SgemmArgs, int, float*, int, int), const SgemmArgs& a, float* scratch, int iters) {
    const int gm = (a.M + MG_BM - 1) / MG_BM, gn = (a.N + MG_BN - 1) / MG_BN;
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(gm * gn, 1, 1), dim3(256), 0, 0, a, a.K, scratch, gn, gm);
    CK(hipEventRecord(t0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(gm * gn, 1, 1), dim3(256), 0, 0, a, a.K, scratch, gn, gm);
    CK(hipEventRecord(t1, 0));
    CK(hipEventSynchronize(t1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, t0, t1));
    return 1e3f * ms / iters;      // us per launch
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 49152, iters = 20;
    const int KMAX = 1536, NMAX = 1536;
    std::vector<unsigned short> h((size_t)M * KMAX);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    bf16_t *A, *B, *c16, *g16; float *C, *bias, *scratch;
    CK(hipMalloc(&A, (size_t)M * KMAX * 2)); CK(hipMalloc(&B, (size_t)NMAX * KMAX * 2));
    CK(hipMalloc(&c16, (size_t)M * NMAX * 2)); CK(hipMalloc(&g16, (size_t)M * NMAX * 2)); CK(hipMalloc(&C, (size_t)M * NMAX * 4));
    CK(hipMalloc(&bias, NMAX * 4)); CK(hipMalloc(&scratch, 1 << 20));
    CK(hipMemcpy(A, h.data(), (size_t)M * KMAX * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)NMAX * KMAX * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, NMAX * 4));
    struct Epi { const char* name; bool c32, c16, g16; };
    const Epi epis[] = {{"no store", false, false, false}, {"bf16", false, true, false}, {"fp32", true, false, false}, {"fp32+bf16", true, true, false},
                        {"bf16+gelu16", false, true, true}};
    printf("| M x N x K | epilogue | 3 wg/CU us | 4 wg/CU us | TFLOP/s (4) | tiles |\n|---|---|---:|---:|---:|---:|\n");
    const int shapes[][2] = {{384, 64}, {384, 128}, {384, 384}, {384, 1536}, {1152, 384}, {1536, 64}, {1536, 128}, {1536, 384}};
    for (const auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        for (const Epi& e : epis) {
            if ((K == 64 || K == 128) && (e.c32 && e.c16)) continue;
            SgemmArgs a{};
            a.A = reinterpret_cast<const float*>(A); a.sam = K; a.sak = 1; a.B = reinterpret_cast<const float*>(B); a.sbk = 1; a.sbn = K;
            a.bias = bias; a.R = nullptr; a.ldr = 0; a.rper = 1; a.C = e.c32 ? C : nullptr; a.ldc = N; a.M = M; a.N = N; a.K = K; a.alpha = 1.f;
            a.a16 = a.b16 = 1; a.c16 = e.c16 ? c16 : nullptr; a.gelu_out16 = e.g16 ? g16 : nullptr;
            const float t3 = run(mfma_bgemm16_kernel<false>, a, scratch, iters), t4 = run(mfma_bgemm16_kernel<true>, a, scratch, iters);
            printf("| %d x %d x %d | %s | %.1f | %.1f | %.0f | %d |\n", M, N, K, e.name, t3, t4, 2.0 * M * N * K / t4 * 1e-6, (M / 128) * (N / 128));
        }
    }
    return 0;
}
