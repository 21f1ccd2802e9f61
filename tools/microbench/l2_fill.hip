// Per-CU fill bandwidth from L2 (shared, hot buffer) into registers / LDS: the streaming roofline of weight-streaming kernels.
// Each workgroup reads the same `bytes`-sized buffer `reps` times.  hipcc --offload-arch=gfx950 -O3 -o l2_fill l2_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void read_regs(const u32x4* __restrict__ buf, size_t n16, int reps, unsigned* sink) {
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < n16; i += UNROLL * 256) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(buf + i + u * 256);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int SLOTS>   // SLOTS x 4 KB (256 threads x 16 B) LDS ring filled by global_load_lds, waited with counted vmcnt
__global__ __launch_bounds__(256) void read_lds(const u32x4* __restrict__ buf, size_t n16, int reps, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[SLOTS * 4096];
    const int wave = threadIdx.x >> 6;
    unsigned x = 0;
    for (int r = 0; r < reps; ++r) {
        const size_t steps = n16 / 256;
        for (size_t s = 0; s < steps; ++s) {
            const int slot = (int)(s % SLOTS);
            __builtin_amdgcn_global_load_lds((const void*)(buf + s * 256 + threadIdx.x),
                                             (__attribute__((address_space(3))) void*)(ring + slot * 4096 + wave * 1024), 16, 0, 0);
            if (slot == SLOTS - 1) { __builtin_amdgcn_s_waitcnt(0x0070 | (0 & 0xF)); x ^= *(volatile unsigned*)(ring + (threadIdx.x & 1023) * 4); }
        }
    }
    if (x == 0x12345678u) sink[0] = 1;
}

int main() {
    const size_t sizes[] = {1u << 20, 2u << 20, 3u << 20, 8u << 20, 64u << 20};
    unsigned* sink; hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (size_t bytes : sizes) {
        u32x4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
        const size_t n16 = bytes / 16;
        for (int blocks : {32, 256, 512, 1024}) {
            const int reps = (int)((256u << 20) / bytes / 4) + 1;
            auto run = [&](auto kern, const char* name) {
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, n16, 1, sink);
                hipEventRecord(a);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, n16, reps, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double per_block = (double)bytes * reps / (ms * 1e-3) / 1e9;
                const int cus = blocks < 256 ? blocks : 256;
                printf("%-14s buf %3zu MB blocks %4d: %7.1f GB/s per block, %8.1f GB/s per CU, %8.2f TB/s total\n", name, bytes >> 20, blocks,
                       per_block, per_block * blocks / cus, per_block * blocks / 1e3);
            };
            run(read_regs<4>, "regs x4");
            run(read_regs<16>, "regs x16");
            run(read_lds<8>, "glds ring8");
            run(read_lds<16>, "glds ring16");
        }
        hipFree(buf);
    }
    return 0;
}
