// gfx950 HBM write rate by store shape: what one wave-wide buffer/global store instruction covers.
//   mode 0: 1 KB contiguous (lane * 16 B)                       mode 1: 8 rows x 128 B, row pitch P
//   mode 2: 16 rows x 64 B, row pitch P                           mode 3: 16 rows x 64 B, two instructions back to back fill 128 B per row
//   mode 4: dword stores, 4 rows x 16 runs of 16 B (the V^T accumulator layout of encoder_blocks.h's record mode)
//   mode 5: mode 2 with 12 instructions back to back covering whole 768-B rows (LayerNorm output rows)
// Every workgroup (256 threads, one per CU x 2) streams its own 2 MB region `reps` times; bytes / time = aggregate write rate.
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned char* out, size_t region, int iters) {
    unsigned char* base = out + (size_t)blockIdx.x * region;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    constexpr int P = 768;      // row pitch in bytes (bf16 [rows][384])
    for (int it = 0; it < iters; ++it) {
        // a "tile" = 128 rows x 768 B = 96 KB; 4 waves x 32 rows
        unsigned char* tile = base + (size_t)(it % (int)(region / (128 * P))) * (128 * P);
        if constexpr (MODE == 0) {
            for (int i = 0; i < 24; ++i) *reinterpret_cast<u32x4*>(tile + (wave * 24 + i) * 1024 + lane * 16) = v;
        } else if constexpr (MODE == 1) {
            for (int j = 0; j < 4; ++j)            // 8 rows x 128 B per instruction; 6 column groups
                for (int c = 0; c < 6; ++c) *reinterpret_cast<u32x4*>(tile + (wave * 32 + j * 8 + (lane >> 3)) * P + c * 128 + (lane & 7) * 16) = v;
        } else if constexpr (MODE == 2) {
            for (int c = 0; c < 12; ++c)           // 16 rows x 64 B per instruction, column-major order: a row's 128-B lines complete late
                for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(tile + (wave * 32 + j * 16 + (lane & 15)) * P + c * 64 + (lane >> 4) * 16) = v;
        } else if constexpr (MODE == 3 || MODE == 5) {
            for (int j = 0; j < 2; ++j)            // 16 rows x 64 B, consecutive instructions fill consecutive 64-B pieces of the same rows
                for (int c = 0; c < 12; ++c) *reinterpret_cast<u32x4*>(tile + (wave * 32 + j * 16 + (lane & 15)) * P + c * 64 + (lane >> 4) * 16) = v;
        } else if constexpr (MODE == 6) {
            for (int j = 0; j < 8; ++j)            // 4 rows x 256 B per instruction; 3 column groups
                for (int c = 0; c < 3; ++c) *reinterpret_cast<u32x4*>(tile + (wave * 32 + j * 4 + (lane >> 4)) * P + c * 256 + (lane & 15) * 16) = v;
        } else if constexpr (MODE == 7) {
            for (int j = 0; j < 2; ++j)            // 16 rows, per row 4 pieces of 16 B at a 32-B stride; the odd pieces by the next instruction
                for (int c = 0; c < 6; ++c)
                    for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<u32x4*>(tile + (wave * 32 + j * 16 + (lane & 15)) * P + c * 128 + (lane >> 4) * 32 + hh * 16) = v;
        } else if constexpr (MODE == 4) {
            // dword stores: lane (rr, g): d = (rr >> 2) * 8 + (rr & 3) (+ 4 (i & 1) + 32 (i >> 1)), token 4 g + r; [rows][192 f32 = 768 B]
            for (int j = 0; j < 2; ++j)
                for (int h = 0; h < 3; ++h)
                    for (int i = 0; i < 4; ++i)
                        for (int r = 0; r < 4; ++r) {
                            const int rr = lane & 15, g = lane >> 4;
                            *reinterpret_cast<unsigned*>(tile + (wave * 32 + j * 16 + 4 * g + r) * P + (h * 64 + (i >> 1) * 32 + (rr >> 2) * 8 + (i & 1) * 4 + (rr & 3)) * 4) = threadIdx.x;
                        }
        }
    }
}
template <int MODE>
void run(const char* what, unsigned char* dev, size_t region, int blocks) {
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dev, region, 20);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dev, region, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * iters * 128 * 768;
    printf("%-92s %7.3f ms  %6.2f TB/s  %6.1f B/clk per workgroup at 2.4 GHz\n", what, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / blocks / 2.4e9);
}
int main() {
    const int blocks = 512; const size_t region = 4 << 20;
    unsigned char* dev; hipMalloc(&dev, region * blocks); hipMemset(dev, 0, region * blocks);
    run<0>("1 KB contiguous per instruction", dev, region, blocks);
    run<1>("8 rows x 128 B per instruction (pitch 768)", dev, region, blocks);
    run<2>("16 rows x 64 B, a row's second 64 B two instructions later", dev, region, blocks);
    run<3>("16 rows x 64 B, consecutive instructions complete the rows", dev, region, blocks);
    run<4>("dword stores in the V^T accumulator layout (runs of 16 B)", dev, region, blocks);
    // fewer workgroups: is ~6 TB/s the chip's write rate or 256 x one CU's?  (rate per CU = TB/s / workgroups resident)
    for (int b : {32, 64, 128, 256}) {
        char what[96]; snprintf(what, sizeof what, "16 rows x 64 B (mode 3), %d workgroups", b);
        run<3>(what, dev, region, b);
    }
    for (int b : {32, 128}) {
        char what[96]; snprintf(what, sizeof what, "1 KB contiguous (mode 0), %d workgroups", b);
        run<0>(what, dev, region, b);
    }
    run<1>("8 rows x 128 B (mode 1), 32 workgroups", dev, region, 32);
    run<2>("16 rows x 64 B, rows completed late (mode 2), 32 workgroups", dev, region, 32);
    run<4>("dword stores, V^T layout (mode 4), 32 workgroups", dev, region, 32);
    run<6>("4 rows x 256 B (mode 6), 32 workgroups", dev, region, 32);
    run<7>("16 rows x 4 pieces of 16 B at 32-B stride (the q | k f32 stores) (mode 7), 32 workgroups", dev, region, 32);
    return 0;
}
