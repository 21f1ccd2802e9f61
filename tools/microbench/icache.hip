// Cost of executing straight-line code ONCE per workgroup (cold instruction cache) vs the same work as a loop.
// Every kernel does N dependent-free FMAs per lane; STRAIGHT = fully unrolled with distinct literal constants
// (~12 bytes of code per FMA), LOOP = 64-instruction body iterated.  hipcc --offload-arch=gfx950 -O3 -o icache icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define ONE { constexpr int c_ = __COUNTER__; x[c_ % 8] = fmaf(x[c_ % 8], 1.0f + 1e-6f * (float)(c_ * 7 % 1013), 0.5f + 1e-6f * (float)(c_ * 3 % 997)); }
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R64(X) R4(R16(X))
#define R256(X) R4(R64(X))
#define R1024(X) R4(R256(X))
template <int N> __device__ __forceinline__ void chain(float (&x)[8]);
template <> __device__ __forceinline__ void chain<64>(float (&x)[8]) { R64(ONE) }
template <> __device__ __forceinline__ void chain<1024>(float (&x)[8]) { R1024(ONE) }
template <> __device__ __forceinline__ void chain<4096>(float (&x)[8]) { R4(R1024(ONE)) }
template <> __device__ __forceinline__ void chain<8192>(float (&x)[8]) { R4(R1024(ONE)) R4(R1024(ONE)) }

template <int N>
__global__ __launch_bounds__(256) void straight(float* out, unsigned long long* cyc) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x * (i + 1);
    const unsigned long long t0 = __builtin_readcyclecounter();
    chain<N>(x);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int N>
__global__ __launch_bounds__(256) void looped(float* out, unsigned long long* cyc) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x * (i + 1);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < N / 64; ++r) chain<64>(x);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
void run(const char* name, F kern, int n, int blocks, float* out, unsigned long long* cyc) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {          // rep 0: cold everything; rep 1: L2 holds the code, I$ state unknown
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[2048]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
        unsigned long long mn = ~0ull, mx = 0, sum = 0;
        for (int i = 0; i < blocks; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; sum += h[i]; }
        printf("%-9s N=%5d blocks=%4d rep%d: kernel %7.1f us; per-block s_memtime ticks (100 MHz) min %6llu avg %6llu max %6llu  -> %.1f ns per FMA (avg)\n",
               name, n, blocks, rep, ms * 1e3, mn, sum / blocks, mx, (double)(sum / blocks) * 10.0 / n);
    }
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 8);
    for (int blocks : {256, 1024}) {
        run("straight", straight<1024>, 1024, blocks, out, cyc);
        run("looped", looped<1024>, 1024, blocks, out, cyc);
        run("straight", straight<4096>, 4096, blocks, out, cyc);
        run("looped", looped<4096>, 4096, blocks, out, cyc);
        run("straight", straight<8192>, 8192, blocks, out, cyc);
        run("looped", looped<8192>, 8192, blocks, out, cyc);
    }
    return 0;
}
