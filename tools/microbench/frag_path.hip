// VERDICT r3 item 3 (i), measured: do the weight fragments of the bf16 one-launch encoder's GEMM phases get to the matrix cores faster
// straight from L2 into VGPRs than through the LDS ring?  The phase's shape, nothing else: one workgroup per CU, four waves of 32 rows
// (two 16-row MFMA tiles), a weight STAGE = 64 output columns x 128 k = 16 KiB = 16 one-KiB fragments; every wave multiplies its rows
// against the WHOLE stage (32 v_mfma_f32_16x16x32_bf16 per wave and stage, each fragment feeding two of them).  All workgroups walk the
// same 3.4 MB of fragment-ordered weights (one block's worth), as the product kernel's workgroups do.
//   MODE 0  fragments fixed in registers: the matrix cores' own time
//   MODE 1  the product's path: each wave copies a quarter of the stage L2 -> LDS (global_load_lds, 16 B per lane), three-slot ring,
//           one counted wait + one workgroup barrier per stage, 16 ds_read_b128 per wave and stage
//   MODE 2  straight into VGPRs: each wave loads all 16 fragments itself (global_load_dwordx4, two register buffers), no LDS, no barrier
//   MODE 3  as 2 but only HALF the stage's fragments come from global memory (the other half stays fixed): what the path costs if fc1
//           alone went direct and fc2 stayed on the ring, priced per direct stage at half traffic
//   MODE 4  the ring of MODE 1 with 64 rows per wave (four row tiles: every fragment feeds four MFMAs, 64 MFMAs per wave and stage) — VERDICT r3
//           item 3 (iii): what halving the fragment reads per MFMA buys the ring path, before the cost of parking x
//   MODE 5  the ring with THREE stages per barrier (six slots), 32 rows per wave: the product kernel's grouping
//   hipcc --offload-arch=gfx950 -O3 -o frag_path tools/microbench/frag_path.hip && ./frag_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int STAGE_BYTES = 16384, FRAGS = 16, STAGES = 216;       // 216 x 16 KiB = 3.4 MB: q|k|v, proj, fc1, fc2 of one block in bf16

__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void walk(const unsigned char* __restrict__ w, const bf16x8* __restrict__ ain, f32x4* __restrict__ out, int passes) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // MODE 1: 3 x 16 KiB
    constexpr int RT = MODE == 4 ? 4 : 2;                 // 16-row tiles per wave
    bf16x8 a[RT][4];
#pragma unroll
    for (int i = 0; i < RT * 4; ++i) a[i >> 2][i & 3] = ain[(i & 7) * 256 + t];
    f32x4 acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT * 4; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage_mma = [&](auto&& frag) {
#pragma unroll
        for (int f = 0; f < FRAGS; ++f) {
            const bf16x8 b = frag(f);
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r][f >> 2] = mma(a[r][f & 3], b, acc[r][f >> 2]);
        }
    };
    const int total = passes * STAGES;
    if constexpr (MODE == 0) {
        bf16x8 fb[FRAGS];
#pragma unroll
        for (int f = 0; f < FRAGS; ++f) fb[f] = *reinterpret_cast<const bf16x8*>(w + f * 1024 + lane * 16);
        for (int s = 0; s < total; ++s) {
            stage_mma([&](int f) { return fb[f]; });
            asm volatile("" ::: "memory");
        }
    } else if constexpr (MODE == 1 || MODE == 4) {
        auto issue = [&](int s) {       // this wave's quarter of stage s: four 1-KiB pieces
            const unsigned char* src = w + (size_t)(s % STAGES) * STAGE_BYTES + wave * 4096 + lane * 16;
            unsigned char* dst = ring + (s % 3) * STAGE_BYTES + wave * 4096;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
        };
        issue(0); issue(1);
        for (int s = 0; s < total; ++s) {
            if (s + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();            // stage s has landed for every wave; slot (s + 2) % 3 has been read by every wave
            if (s + 2 < total) issue(s + 2);
            const unsigned char* st = ring + (s % 3) * STAGE_BYTES + lane * 16;
            stage_mma([&](int f) { return *reinterpret_cast<const bf16x8*>(st + f * 1024); });
        }
    } else if constexpr (MODE == 5) {
        auto issue3 = [&](int gidx) {   // this wave's quarter of the three stages of group gidx: twelve 1-KiB pieces
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int s = gidx * 3 + q;
                const unsigned char* src = w + (size_t)(s % STAGES) * STAGE_BYTES + wave * 4096 + lane * 16;
                unsigned char* dst = ring + (s % 6) * STAGE_BYTES + wave * 4096;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 1024),
                                                     (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
            }
        };
        const int groups = total / 3;
        issue3(0);
        for (int gi = 0; gi < groups; ++gi) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();            // group gi has landed for every wave; the other three slots have been read by every wave
            if (gi + 1 < groups) issue3(gi + 1);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned char* st = ring + ((gi * 3 + q) % 6) * STAGE_BYTES + lane * 16;
                stage_mma([&](int f) { return *reinterpret_cast<const bf16x8*>(st + f * 1024); });
            }
        }
    } else {
        constexpr int NL = MODE == 2 ? FRAGS : FRAGS / 2;      // fragments per stage that come from global memory
        bf16x8 fb[2][FRAGS];
#pragma unroll
        for (int f = 0; f < FRAGS; ++f) fb[0][f] = fb[1][f] = *reinterpret_cast<const bf16x8*>(w + f * 1024 + lane * 16);
        auto load = [&](int s, bf16x8 (&dst)[FRAGS]) {
            const unsigned char* src = w + (size_t)(s % STAGES) * STAGE_BYTES + lane * 16;
#pragma unroll
            for (int f = 0; f < NL; ++f) dst[f] = *reinterpret_cast<const bf16x8*>(src + f * 1024);
        };
        load(0, fb[0]);
        for (int s = 0; s < total; s += 2) {
            load(s + 1, fb[1]);
            stage_mma([&](int f) { return fb[0][f]; });
            load(s + 2, fb[0]);
            stage_mma([&](int f) { return fb[1][f]; });
        }
    }
#pragma unroll
    for (int i = 0; i < RT * 4; ++i) out[(size_t)(blockIdx.x * 16 + i) * 256 + t] = acc[i >> 2][i & 3];
}

template <int MODE>
static int run(const char* name, const unsigned char* w, const bf16x8* a, f32x4* out, int passes, double mhz, double base_us) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t lds = (MODE == 1 || MODE == 4) ? 3 * STAGE_BYTES : MODE == 5 ? 6 * STAGE_BYTES : 0;
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(walk<MODE>, dim3(256), dim3(256), lds, 0, w, a, out, passes);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double us = best * 1e3, per_stage_clk = us * mhz / ((double)passes * STAGES);
    printf("%-46s %9.1f us   %7.1f clk per stage   %5.1f %% of it is the 32 MFMAs' own time\n", name, us, per_stage_clk,
           base_us > 0 ? 100.0 * base_us / us : 100.0);
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const double mhz = prop.clockRate / 1e3;
    printf("%s, %d CUs, %.0f MHz; 256 workgroups x 4 waves, stage = 16 KiB, %d stages per pass\n", prop.gcnArchName, prop.multiProcessorCount, mhz, STAGES);
    const size_t wbytes = (size_t)(STAGES + 3) * STAGE_BYTES;
    unsigned char* w; bf16x8* a; f32x4* out;
    CHECK(hipMalloc(&w, wbytes)); CHECK(hipMalloc(&a, 8 * 256 * sizeof(bf16x8))); CHECK(hipMalloc(&out, (size_t)256 * 16 * 256 * sizeof(f32x4)));
    std::vector<unsigned short> h(wbytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 25);     // bf16 values near 0.01
    CHECK(hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(a, h.data(), 8 * 256 * sizeof(bf16x8), hipMemcpyHostToDevice));
    const int passes = 12;          // twelve blocks
    hipLaunchKernelGGL(walk<0>, dim3(256), dim3(256), 0, 0, w, a, out, passes); CHECK(hipDeviceSynchronize());
    // base: MODE 0
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double base = 1e30;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(walk<0>, dim3(256), dim3(256), 0, 0, w, a, out, passes);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3 < base) base = ms * 1e3;
    }
    if (run<0>("0 fragments fixed in registers", w, a, out, passes, mhz, base)) return 1;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(walk<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES));
    if (run<1>("1 L2 -> LDS ring (product path)", w, a, out, passes, mhz, base)) return 1;
    if (run<2>("2 L2 -> VGPRs, every wave the whole stage", w, a, out, passes, mhz, base)) return 1;
    if (run<3>("3 L2 -> VGPRs, half of every stage", w, a, out, passes, mhz, base)) return 1;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(walk<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * STAGE_BYTES));
    if (run<5>("5 ring, three stages per barrier", w, a, out, passes, mhz, base)) return 1;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(walk<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES));
    if (run<4>("4 ring, 64 rows per wave (2x the MFMAs per stage)", w, a, out, passes, mhz, 2.0 * base)) return 1;
    return 0;
}
