// Common device helpers for the PARSeq gfx950 kernels: storage types, 16-byte fragments, MFMA wrappers.
// Written for CDNA4 only (wave64, v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x4_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace pq {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // a 16-byte chunk in registers (HIP's uint4 is a struct:
                                                                   // selects on it go through scratch memory)

constexpr int WAVE = 64;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return static_cast<float>(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return static_cast<bf16_t>(v); }  // RNE

// A 16-byte operand fragment: 8 bf16 or 4 f32, exactly what one lane feeds to one k-group of a 16x16 (or 32x32)
// MFMA.  In BYTES the two storage types look the same (64 bytes of K per 16x16 k-group), which lets the tiled
// kernels share all their address arithmetic.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { f32x4 v; };

template <typename T> constexpr int frag_elems() { return 16 / (int)sizeof(T); }

// acc(16x16 tile, 4 f32 per lane) += A(16 x K-group) * B(K-group x 16).
// Operand convention (cdna_hip_programming.md section 3): the first operand's lane l supplies row (l & 15),
// k-slots of group (l >> 4); the second operand's lane l supplies column (l & 15), same k-slots.
// Result: lane l holds D[row = 4 * (l >> 4) + r][col = l & 15], r = 0..3.
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
    // exact-f32 MFMA: 4 instructions of K=4, lane group g = l >> 4 supplies k = g for each; element j of the
    // 16-byte fragment is k-slot (4 g + j) of the 16-wide k-group.  Bitwise an fmaf chain.
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[0], b.v[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[1], b.v[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[2], b.v[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[3], b.v[3], acc, 0, 0, 0);
}

// Wavefront reductions on the VALU's DPP lanes (no LDS round trips: a __shfl_xor butterfly compiles to six dependent
// ds_bpermute + s_waitcnt pairs, ~600 cycles per reduction; this is four DPP adds + four readlanes).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// every lane ends with the sum / max over its aligned row of 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// All-reduce over the four 16-lane rows of a wavefront (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: gfx950's row / half swaps
// exchange whole 16- / 32-lane groups between two registers, so with both operands the same value the two results are
// "my group's" and "the other group's" copy.  (__shfl_xor(v, 16) compiles to ds_bpermute: an LDS-pipe round trip plus an index
// register per distance that the compiler keeps live across whole kernels.)
//   v_permlane16_swap: odd rows of the first operand <-> even rows of the second;  v_permlane32_swap: upper half <-> lower half.
__device__ __forceinline__ float rows4_sum(float v) {
    const unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float w = __uint_as_float(a[0]) + __uint_as_float(a[1]);          // rows (0,1) and (2,3) summed pairwise, in every lane of the pair
    const unsigned x = __float_as_uint(w);
    auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows4_max(float v) {
    const unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float w = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned x = __float_as_uint(w);
    auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// erf(x) to ~1.2e-7 absolute (Abramowitz & Stegun 7.1.26 with the 5-term polynomial, evaluated in fp32): one
// reciprocal, one exp, a Horner chain — about 15 VALU ops instead of the ~100 of the library erff with its branches,
// which alone cost the fc1 epilogue ~100 us per launch at 100 M activations.  Error budget: |d gelu| <= 0.5 |x| 1.2e-7.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp): `1.0f / x` expands to the ~10-instruction IEEE division
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}

#ifdef PQ_GELU_AS_WRITTEN      // the A/B of round 6: 0.5 x (1 + erf(x / sqrt 2)) through fast_erf, 16 VALU instructions per value
__device__ __forceinline__ float gelu_erf(float x) {
    // exact-erf GELU (torch.nn.GELU() default / F.gelu): 0.5 x (1 + erf(x / sqrt(2)))
    return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f));
}
#else
// exact-erf GELU (torch.nn.GELU() default / F.gelu), the same A & S 7.1.26 erf with the algebra done before the arithmetic (round 6: the
// one-launch bf16x3 encoder is bound by the instructions it issues beside its MFMAs, and the GELU is one per MFMA of the MLP phase):
//   0.5 x (1 + erf(x / sqrt 2)) = max(x, 0) - |x| q(|x|),   q(u) = 0.5 P(t) t exp(-u^2 / 2),   t = 1 / (1 + (0.3275911 / sqrt 2) u)
// — the 1 / sqrt 2 lives in the reciprocal's constant, the 0.5 in the polynomial's coefficients (exact: a power of two), exp(-u^2 / 2) is
// one v_exp_f32 of -(u sqrt(0.5 log2 e))^2, and sign handling is the max: 13 VALU instructions per value (two of them transcendental)
// instead of 16.  For x < 0 there is no 1 - (1 - small) cancellation any more: |error| <= 0.5 |x| 1.2e-7 as before, smaller in the left tail.
__device__ __forceinline__ float gelu_erf(float x) {
    const float u = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, u, 1.0f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float s = u * 0.84932180028801904272f;                     // sqrt(0.5 log2 e)
    const float e = __builtin_amdgcn_exp2f(-(s * s));
    return fmaf(-u, p * t * e, __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()));      // med3(x, 0, inf) = max(x, 0) in one instruction
}
#endif

// GELU for bf16-stored outputs: erf(z) = z * P(z^2) on |z| <= 3.5 (clamped; erf(3.5) = 1 - 7e-7), P of degree 9 from Chebyshev
// interpolation — 13 full-rate FMAs / MULs that pack into v_pk_fma_f32, no transcendental.  |erf error| <= 7.2e-5,
// |gelu error| <= 1.8e-4 absolute (<= 4e-5 relative where it peaks): an order of magnitude inside the bf16 rounding that
// follows (2^-9 relative).  The exact-mode (f32) path keeps gelu_erf.  (PMC: VALU was 33 % of the fused MLP kernel's wave
// cycles, 4.3 cycles per VALU instruction, two quarter-rate transcendentals per element.)
__device__ __forceinline__ float gelu_poly(float x) {
    const float z = fminf(fmaxf(x * 0.70710678118654752440f, -3.5f), 3.5f);
    const float t = z * z;
    float p = -1.417363337807842e-09f;
    p = fmaf(p, t, 9.542367251924588e-08f);
    p = fmaf(p, t, -2.8274080250412226e-06f);
    p = fmaf(p, t, 4.8881945986067876e-05f);
    p = fmaf(p, t, -0.0005533255753107369f);
    p = fmaf(p, t, 0.004382880870252848f);
    p = fmaf(p, t, -0.025440679863095284f);
    p = fmaf(p, t, 0.11156132817268372f);
    p = fmaf(p, t, -0.3756689131259918f);
    p = fmaf(p, t, 1.1283513307571411f);
    const float hx = 0.5f * x;
    return fmaf(hx, p * z, hx);
}
// gelu_poly's polynomial over 16 values with the polynomial steps OUTERMOST: every step is eight independent v_pk_fma_f32, so no
// instruction depends on the one before it.  Evaluated one element at a time the compiler emits each Horner chain as ten
// dependent packed FMAs with an s_nop after every one of them (the wait state a dependent packed pair needs) — in a kernel with one
// wave per SIMD those are issue slots nothing else can fill.  The empty asm between steps ties all eight accumulators to one
// program point: instruction selection linearises pure arithmetic by chain again otherwise (sched_barrier only binds the later
// machine scheduler).  The 1/sqrt(2) argument scaling and the final 0.5 are folded into the coefficients:
//     gelu(x) = x (0.5 + c P'(c^2)),  c = clamp(x, +-3.5 sqrt 2),  P'_i = P_i / (2 sqrt 2 * 2^i)
// — 14 packed operations per pair instead of 17; same polynomial, so the same 2e-4 absolute bound (not bit-identical to gelu_poly:
// the coefficient products round differently).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_poly_16(const f32x2 (&x)[8], f32x2 (&y)[8]) {
    f32x2 z[8], t[8], p[8];
    constexpr float kLim = 4.949747468305833f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        z[e] = __builtin_elementwise_min(__builtin_elementwise_max(x[e], f32x2{-kLim, -kLim}), f32x2{kLim, kLim});
        t[e] = z[e] * z[e];
        p[e] = __builtin_elementwise_fma(f32x2{-9.78737526922973e-13f, -9.78737526922973e-13f}, t[e], f32x2{1.3178657407047493e-10f, 1.3178657407047493e-10f});
    }
    constexpr float kC[8] = {-7.809685108155907e-09f, 2.700371522214307e-07f, -6.113441664158902e-06f, 9.684889951526828e-05f,
                             -0.00112432982807442f, 0.00986072145863531f, -0.0664095089880922f, 0.3989324387696197f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = __builtin_elementwise_fma(p[e], t[e], f32x2{kC[s], kC[s]});
    }
    asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = x[e] * __builtin_elementwise_fma(z[e], p[e], f32x2{0.5f, 0.5f});
}
template <typename T> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<bf16_t>(float x) { return gelu_poly(x); }

template <typename T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
    bf16x4 o;
    o[0] = static_cast<bf16_t>(v[0]); o[1] = static_cast<bf16_t>(v[1]);
    o[2] = static_cast<bf16_t>(v[2]); o[3] = static_cast<bf16_t>(v[3]);
    *reinterpret_cast<bf16x4*>(p) = o;
}

// Greedy pick with torch.argmax's semantics, safe for any input: the first maximum wins, a NaN counts as the maximum (the first
// NaN wins), and a row with no finite winner (all -inf) yields index 0 — never the "no candidate yet" sentinel, which would be
// used as a token id to index the decoder tables.
constexpr int ARGMAX_NONE = 0x7fffffff;
__device__ __forceinline__ bool argmax_take(float v, int i, float best, int bi) {
    // bitwise on purpose: the short-circuit form compiles to a tree of divergent branches (27 of them per wave-wide arg-max)
    const bool vn = v != v, bn = best != best, lt = i < bi;
    return (vn & (!bn | lt)) | (!vn & !bn & ((v > best) | ((v == best) & lt)));
}
__device__ __forceinline__ int argmax_final(int bi, int C) { return bi < C ? bi : 0; }
// Wave-wide arg-max of (best, bi) pairs under argmax_take's total order (so the combining order does not matter): four DPP steps make
// every row of 16 lanes uniform, four readlanes combine the rows.  Every lane ends with the winner.  (The __shfl_xor butterfly it replaces
// is twelve dependent ds_bpermute round trips: ~2 us of the AR step's pick, tools/ds_step_timers.py.)
template <int CTRL>
__device__ __forceinline__ void argmax_dpp_step(float& best, int& bi) {
    const float ov = dpp_mov<CTRL>(best);
    const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xF, 0xF, true);
    if (argmax_take(ov, oi, best, bi)) { best = ov; bi = oi; }
}
__device__ __forceinline__ void wave_argmax(float& best, int& bi) {
    argmax_dpp_step<0xB1>(best, bi);
    argmax_dpp_step<0x4E>(best, bi);
    argmax_dpp_step<0x141>(best, bi);
    argmax_dpp_step<0x140>(best, bi);
    float b = lane_bcast(best, 0); int i = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float ov = lane_bcast(best, r);
        const int oi = __builtin_amdgcn_readlane(bi, r);
        if (argmax_take(ov, oi, b, i)) { b = ov; i = oi; }
    }
    best = b; bi = i;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel.  One LdsAttr per launch site remembers
// (bit per device ordinal, lock-free) on which devices the attribute has been raised, so that a process driving several GPUs
// — or several host threads, one plan each — sets it wherever it launches; setting it twice is harmless.
struct LdsAttr {
    std::atomic<uint64_t> done{0};
    hipError_t ensure(const void* fn, size_t bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
        return e;
    }
};

}  // namespace pq
