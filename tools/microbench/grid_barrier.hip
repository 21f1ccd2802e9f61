// VERDICT r5 item 4, measured instead of estimated: what would ONE cooperative launch of the 26-step AR loop pay per grid-wide
// synchronisation on an MI355X, against the dependent kernel boundary the loop pays today (3 launches per step)?
//
//   1. barrier-counter: one monotonic device-scope counter (lane 0: release fence, relaxed agent-scope add, relaxed sc1 poll + s_sleep, acquire fence)
//   2. barrier-xcd:     per-XCD arrival counter (XCC id from the hardware register, group sizes from a census), the XCD's last arriver releases,
//                       arrives at the top counter, waits for all XCDs, acquires and bumps the XCD's generation word; the others poll that and acquire
//   3. the same two with the AR step's activation exchange between barriers: every workgroup publishes its 3 KiB slice of a 512 x 384 f32
//      activation matrix (786 KB in all), and after the barrier reads the 24 KiB (16 rows x 384) a (row tile, column tile) product would consume
//   4. the exchange as a kernel boundary: the same publish / consume body as a chain of plain launches on one stream
//
// Prints us per barrier (kernel time of N barriers minus the kernel with none, / N) and us per launch of the chain.  One workgroup per CU (256 x 256
// threads and 256 x 512 threads).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/grid_barrier tools/microbench/grid_barrier.hip && ./tools/microbench/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Sync {
    unsigned top;            // arrivals of XCD leaders (barrier-xcd) or of every workgroup (barrier-counter)
    unsigned pad0[31];
    unsigned xcc_arrive[8 * 32];   // one 128-byte line per XCD
    unsigned xcc_gen[8 * 32];
    unsigned xcc_size[8 * 32];
    unsigned census;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void spin_until(const unsigned* p, unsigned target, long long* guard) {
    while ((int)(ld_relaxed(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++*guard > (1ll << 26)) { printf("grid barrier spin bound hit (block %d)\n", (int)blockIdx.x); __builtin_trap(); }
    }
}

// barrier-counter: epoch e (1, 2, ...) completes when top == e * gridDim.x
__device__ __forceinline__ void barrier_counter(Sync* s, unsigned epoch, long long* guard) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(&s->top, epoch * gridDim.x, guard);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// barrier-xcd: nx = number of XCDs that hold workgroups of this grid (census)
__device__ __forceinline__ void barrier_xcd(Sync* s, unsigned epoch, unsigned x, unsigned xsize, unsigned nx, long long* guard) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's stores have reached its XCD's L2
        const unsigned t = __hip_atomic_fetch_add(&s->xcc_arrive[32 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == epoch * xsize - 1) {                          // the XCD's last arriver: publish the XCD's L2, meet the other leaders, take theirs
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until(&s->top, epoch * nx, guard);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&s->xcc_gen[32 * x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            spin_until(&s->xcc_gen[32 * x], epoch, guard);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

// publish: workgroup w owns row tile w / 8 (16 rows) and column slice w % 8 (48 of 384 columns): 768 floats, 3 KiB
__device__ __forceinline__ void publish(float* act, float v) {
    const unsigned w = blockIdx.x, rt = w >> 3, cs = w & 7;
    for (unsigned i = threadIdx.x; i < 768; i += blockDim.x) {
        const unsigned r = i / 48, c = i % 48;
        act[(size_t)(16 * rt + r) * 384 + 48 * cs + c] = v + (float)i;
    }
}
// consume: the 16 x 384 rows of this workgroup's row tile (24 KiB written by eight workgroups)
__device__ __forceinline__ float consume(const float* act) {
    const unsigned rt = blockIdx.x >> 3;
    const float4* p = reinterpret_cast<const float4*>(act + (size_t)16 * rt * 384);
    float acc = 0.f;
    for (unsigned i = threadIdx.x; i < 1536; i += blockDim.x) { const float4 q = p[i]; acc += q.x + q.y + q.z + q.w; }
    return acc;
}

template <int KIND, bool EXCHANGE>      // KIND 0 none, 1 counter, 2 xcd
__global__ void persistent(Sync* s, float* act0, float* act1, float* out, int iters) {
    long long guard = 0;
    unsigned x = 0, xsize = 0, nx = 0;
    if (KIND == 2) {
        __shared__ unsigned sh[3];
        if (threadIdx.x == 0) {
            x = xcc_id();
            __hip_atomic_fetch_add(&s->xcc_size[32 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&s->census, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until(&s->census, gridDim.x, &guard);
            unsigned n = 0;
            for (int k = 0; k < 8; ++k) n += ld_relaxed(&s->xcc_size[32 * k]) != 0;
            sh[0] = x; sh[1] = ld_relaxed(&s->xcc_size[32 * x]); sh[2] = n;
        }
        __syncthreads();
        x = sh[0]; xsize = sh[1]; nx = sh[2];
    }
    float v = (float)blockIdx.x;
    for (int it = 1; it <= iters; ++it) {
        float* wr = (it & 1) ? act0 : act1;
        if (EXCHANGE) publish(wr, v);
        if (KIND == 1) barrier_counter(s, (unsigned)it, &guard);
        if (KIND == 2) barrier_xcd(s, (unsigned)it, x, xsize, nx, &guard);
        if (EXCHANGE) v = consume(wr) * 1e-9f;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

__global__ void boundary_body(float* wr, const float* rd, float* out) {
    float v = consume(rd) * 1e-9f;
    publish(wr, v);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

template <int KIND, bool EXCHANGE>
static float run(int threads, int iters, Sync* s, float* a0, float* a1, float* out, bool check = false) {
    CK(hipMemset(s, 0, sizeof(Sync)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((persistent<KIND, EXCHANGE>), dim3(256), dim3(threads), 0, 0, s, a0, a1, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (check && EXCHANGE) {
        // after `iters` rounds every workgroup's value is a pure function of the exchange: all workgroups of a row tile must agree
        std::vector<float> h(256);
        CK(hipMemcpy(h.data(), out, 256 * sizeof(float), hipMemcpyDeviceToHost));
        for (int w = 0; w < 256; ++w) if (h[w] != h[w & ~7]) { printf("EXCHANGE MISMATCH kind %d block %d: %g vs %g\n", KIND, w, h[w], h[w & ~7]); break; }
    }
    return ms * 1e3f;
}

int main() {
    Sync* s; float *a0, *a1, *out;
    CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&a0, 512 * 384 * 4)); CK(hipMalloc(&a1, 512 * 384 * 4)); CK(hipMalloc(&out, 256 * 4));
    CK(hipMemset(a0, 0, 512 * 384 * 4)); CK(hipMemset(a1, 0, 512 * 384 * 4));
    const int N = 2000;
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {      // rep 0 warms clocks and code
            const float none = run<0, false>(threads, N, s, a0, a1, out), none_x = run<0, true>(threads, N, s, a0, a1, out);
            const float c = run<1, false>(threads, N, s, a0, a1, out), cx = run<1, true>(threads, N, s, a0, a1, out, true);
            const float x = run<2, false>(threads, N, s, a0, a1, out), xx = run<2, true>(threads, N, s, a0, a1, out, true);
            if (rep)
                printf("256 workgroups x %d threads, %d rounds: barrier-counter %.2f us, barrier-xcd %.2f us per barrier (nothing published); "
                       "with the 786 KB activation exchange: publish + consume alone %.2f us per round, + barrier-counter %.2f us, + barrier-xcd %.2f us per round\n",
                       threads, N, (c - none) / N, (x - none) / N, none_x / N, cx / N, xx / N);
        }
    }
    // the same exchange as a chain of dependent launches (what the AR loop pays today, three times per step)
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int it = 0; it < N; ++it) hipLaunchKernelGGL(boundary_body, dim3(256), dim3(threads), 0, 0, (it & 1) ? a0 : a1, (it & 1) ? a1 : a0, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("chain of %d dependent launches of the same publish + consume body, 256 x %d threads: %.2f us per launch\n", N, threads, ms * 1e3f / N);
        }
    }
    return 0;
}
