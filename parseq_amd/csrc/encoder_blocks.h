// The whole ViT encoder stack as ONE persistent launch (bf16 throughput mode, E = 384, 128 tokens = one image per workgroup):
//
//     for every block:   x += proj(attention(qkv(LayerNorm1(x))));   x += fc2(gelu(fc1(LayerNorm2(x))))          (timm Block x depth)
//
// encoder_attn_fused.h and encoder_mlp.h (RESIDENT form) each keep the fp32 rows of x in the accumulators of their last GEMM
// from the first load to the final store.  A workgroup owns the same 128 rows — one image — in both, and nothing but x crosses
// from one to the other, so the two kernel bodies chain without x ever leaving the register file: this kernel loads x once,
// walks all `depth` blocks, and stores x once.  Per forward at batch 512 that removes 2 x 12 launches' worth of prologue /
// epilogue HBM bursts (x: 24 x 200 MB -> 200 MB) and every one of the 24 launch-wide synchronisation points at which the
// slowest workgroup of a launch used to hold back the next launch.
//
// The two phases are device functions shared with two stand-alone branch kernels in this file (attn_branch_kernel,
// mlp_branch_kernel), whose per-kernel parity tests therefore cover them, and against a chain of which the one-launch encoder is
// tested bit for bit.  (encoder_attn_fused.h / encoder_mlp.h hold the first stand-alone forms, with their ablation variants.)
//   attn_phase   24 weight triples (6 heads x {q, k, v, proj}) through ring groups 0-1, K / V^T images   (encoder_attn_fused.h)
//   mlp_phase    48 weight triples (24 hidden chunks x {fc1, fc2}) through ring groups 0-2               (encoder_mlp.h)
// LDS map (bytes): [0, 96 K) ring groups 0-1 | [96 K, 144 K) ring group 2 of the MLP phase, overlaid by the K and V^T images of
// the attention phase (dead while the other phase runs; the barriers of either phase separate the last reads of one use from
// the first writes of the other) | [144 K, +10.5 K) the phase's biases and LayerNorm parameters.
#pragma once
#include "common.h"
#include "encoder_attn_fused.h"
#include "encoder_mlp.h"
#include "encoder_panel.h"

namespace pq {

// (The timing ablations of this kernel — GELU / soft-max core / LDS-DMA issue / barriers / fragment reads / DMA waits removed one at
// a time, profiles/r02_enc_ablation.log — and the rejected scheduling variants are in the history at commit c342b0c (the parent of the pruning commit c0ffb4f).)

// Per-block parameters, one entry per encoder block: ELEMENT offsets relative to two bases the kernel receives once — the bf16
// weights (wqkv, wproj, w1, w2) relative to `wbase`, the fp32 vectors relative to `pbase`.  32-bit offsets (instead of twelve
// 64-bit pointers) keep the scalar register file free, and the weights are addressed through ONE buffer descriptor.
struct EncBlockParams {
    unsigned ln1_w, ln1_b, wqkv, bqkv, wproj, bproj, ln2_w, ln2_b, w1, b1, w2, b2;
};
// Optional tail of the one-launch encoder (parseq_forward: nobody asked for `memory` itself): the encoder's final LayerNorm and the
// decoder's cross-attention K / V projection of the result, straight from the resident rows — the final LayerNorm launch, the
// bf16 copy of memory, the K / V GEMM launch and the store of x all disappear.  Element offsets like EncBlockParams; wkv / bkv point
// at the K rows of in_proj_weight / in_proj_bias (the K | V part, 2E rows); kmem / vmem: [B][heads][128][hd] bf16, hd = 32.
struct EncTailParams {
    unsigned norm_w, norm_b, wkv, bkv;
    bf16_t* kmem; bf16_t* vmem;          // kmem == nullptr: no tail, x is stored instead
    int heads;
};

constexpr int EB_RING_BYTES = 9 * 16384;      // three groups of three 16 KiB weight stages; group 2 is overlaid by the K / V^T images
template <int E>
constexpr size_t enc_blocks_lds() { return (size_t)EB_RING_BYTES + (size_t)(7 * E) * sizeof(float); }

// ---- x <-> accumulators --------------------------------------------------------------------------------------------------
// Accumulator layout (encoder_mlp.h): lane (r16, g), row tile j, tile pair q32 = (acc[(q32 >> 2) * 8 + 2 (q32 & 3)], [... + 1]) holds
// columns 32 q32 + 8 g + [0, 8) of row 32 wid + 16 j + r16 — also the MFMA operand-fragment layout of k-step q32.
template <int E>
__device__ __forceinline__ void load_x_to_acc(const float* __restrict__ x, int m0, int M, int wid, int rr, int g, f32x4 (&acc)[E / 16][2]) {
    constexpr int KSTEPS = E / 32;
    const bool lo_half = rr < 8;
    u32x4 raw0[2][KSTEPS], raw1[2][KSTEPS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rbase = m0 + wid * 32 + j * 16 + (rr & 7);
        const float* xlo = x + (size_t)min(rbase, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
        const float* xhi = x + (size_t)min(rbase + 8, M - 1) * E + 8 * g + (lo_half ? 0 : 4);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            raw0[j][ks] = *reinterpret_cast<const u32x4*>(xlo + ks * 32);        // a piece of row (r16 & 7)
            raw1[j][ks] = *reinterpret_cast<const u32x4*>(xhi + ks * 32);        // a piece of row (r16 & 7) + 8
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const u32x4 p0 = raw0[j][ks], p1 = raw1[j][ks];
            const u32x4 got = swap_half_rows(lo_half ? p1 : p0);
            const u32x4 ev = lo_half ? p0 : got, od = lo_half ? got : p1;
            acc[(ks >> 2) * 8 + 2 * (ks & 3)][j] = f32x4{__uint_as_float(ev[0]), __uint_as_float(ev[1]), __uint_as_float(ev[2]), __uint_as_float(ev[3])};
            acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j] = f32x4{__uint_as_float(od[0]), __uint_as_float(od[1]), __uint_as_float(od[2]), __uint_as_float(od[3])};
        }
}

template <int E>
__device__ __forceinline__ void store_acc_to_x(float* __restrict__ x, int m0, int M, int wid, int rr, int g, const f32x4 (&acc)[E / 16][2]) {
    const bool lo_half = rr < 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int mrow = m0 + wid * 32 + j * 16;
        const int r_first = mrow + (rr & 7), r_second = r_first + 8;
        const int cbase = 8 * g + (lo_half ? 0 : 4);
#pragma unroll
        for (int q32 = 0; q32 < E / 32; ++q32) {
            const f32x4 ta = acc[(q32 >> 2) * 8 + 2 * (q32 & 3)][j], tb = acc[(q32 >> 2) * 8 + 2 * (q32 & 3) + 1][j];
            const u32x4 pa = {__float_as_uint(ta[0]), __float_as_uint(ta[1]), __float_as_uint(ta[2]), __float_as_uint(ta[3])};
            const u32x4 pb = {__float_as_uint(tb[0]), __float_as_uint(tb[1]), __float_as_uint(tb[2]), __float_as_uint(tb[3])};
            const u32x4 got = swap_half_rows(lo_half ? pb : pa);
            const u32x4 first = lo_half ? pa : got, second = lo_half ? got : pb;
            const int col = 32 * q32 + cbase;
            if (r_first < M) *reinterpret_cast<u32x4*>(x + (size_t)r_first * E + col) = first;
            if (r_second < M) *reinterpret_cast<u32x4*>(x + (size_t)r_second * E + col) = second;
        }
    }
}

// acc += bias[column] (bias in LDS, [E])
template <int E>
__device__ __forceinline__ void add_bias_to_acc(const float* sb, int g, f32x4 (&acc)[E / 16][2]) {
#pragma unroll
    for (int q32 = 0; q32 < E / 32; ++q32) {
        const float4 b0 = *reinterpret_cast<const float4*>(sb + 32 * q32 + 8 * g), b1 = *reinterpret_cast<const float4*>(sb + 32 * q32 + 8 * g + 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4& ta = acc[(q32 >> 2) * 8 + 2 * (q32 & 3)][j];
            f32x4& tb = acc[(q32 >> 2) * 8 + 2 * (q32 & 3) + 1][j];
            ta[0] += b0.x; ta[1] += b0.y; ta[2] += b0.z; ta[3] += b0.w;
            tb[0] += b1.x; tb[1] += b1.y; tb[2] += b1.z; tb[3] += b1.w;
        }
    }
}

// LayerNorm of the rows held in the accumulators -> bf16 operand fragments (statistics: own 96 values + the three lanes r16 + 16 k)
template <int E>
__device__ __forceinline__ void ln_acc_to_frag(const f32x4 (&acc)[E / 16][2], const float* sgam, const float* sbet, float eps, int g,
                                               bf16x8 (&afrag)[2][E / 32]) {
    constexpr int KSTEPS = E / 32;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float s1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const f32x4 a = acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], b = acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j];
            s1 += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
        }
        s1 = rows4_sum(s1);
        const float mean = s1 * (1.0f / E);
        float s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const f32x4 a = acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], b = acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j];
            const float d0 = a[0] - mean, d1 = a[1] - mean, d2 = a[2] - mean, d3 = a[3] - mean;
            const float d4 = b[0] - mean, d5 = b[1] - mean, d6 = b[2] - mean, d7 = b[3] - mean;
            s2 += ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
        }
        s2 = rows4_sum(s2);
        const float rstd = __builtin_amdgcn_rsqf(s2 * (1.0f / E) + eps);       // v_rsq_f32, 1 ulp
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const f32x4 a = acc[(ks >> 2) * 8 + 2 * (ks & 3)][j], b = acc[(ks >> 2) * 8 + 2 * (ks & 3) + 1][j];
            const float4 ga = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g), gb = *reinterpret_cast<const float4*>(sgam + ks * 32 + 8 * g + 4);
            const float4 ba = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g), bb = *reinterpret_cast<const float4*>(sbet + ks * 32 + 8 * g + 4);
            bf16x8 f;
            f[0] = static_cast<bf16_t>((a[0] - mean) * rstd * ga.x + ba.x); f[1] = static_cast<bf16_t>((a[1] - mean) * rstd * ga.y + ba.y);
            f[2] = static_cast<bf16_t>((a[2] - mean) * rstd * ga.z + ba.z); f[3] = static_cast<bf16_t>((a[3] - mean) * rstd * ga.w + ba.w);
            f[4] = static_cast<bf16_t>((b[0] - mean) * rstd * gb.x + bb.x); f[5] = static_cast<bf16_t>((b[1] - mean) * rstd * gb.y + bb.y);
            f[6] = static_cast<bf16_t>((b[2] - mean) * rstd * gb.z + bb.z); f[7] = static_cast<bf16_t>((b[3] - mean) * rstd * gb.w + bb.w);
            afrag[j][ks] = f;
        }
    }
}

// The lane id through an instruction the optimiser cannot see through: values derived from it are NOT loop-invariant to the
// compiler, so the per-lane LDS offsets of a stage are recomputed (half a dozen VALU operations) where they are used instead of being
// hoisted out of the head / chunk loops, kept live across them and — at 512 registers — spilled, which costs far more than the
// arithmetic: a scratch reload inside the loop carries an s_waitcnt vmcnt(0) that also drains the LDS-DMA ring.
__device__ __forceinline__ int opaque_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// ---- record mode (the training forward, lib_train.hip parseq_train_encoder_forward) ------------------------------------------------
// The same walk over the blocks, which additionally WRITES what the hand-derived backward reads: per block the input rows x (f32), the
// two LayerNorm outputs n1 / n2 (bf16), q | k | v with their biases (f32, [rows][3E]), the attention output (bf16), x after the
// attention branch (f32), the fc1 pre-activation and its GELU (bf16, [rows][4E]) — the slots of TrainEncoderLayout, rows in token
// order.  Every store goes through ONE buffer descriptor per block (a block's record spans < 4 GB): a 32-bit per-lane offset, the
// uniform part in a scalar register — no 64-bit per-lane pointers in a kernel that has no registers to spare.  The stores count in
// vmcnt beside the LDS-DMA pieces; the phases' waits stay correct as they are (vmcnt <= N leaves at most N LOADS outstanding, loads
// complete in order, so everything older than the newest N loads has landed) and merely also wait for older stores.
struct EncRecordParams {
    float* base;                 // block 0's record; block l's starts layer_stride floats further
    size_t layer_stride;
    unsigned layer_bytes;        // bytes a block's record spans (the descriptor's range)
    unsigned qkv, ao, x_mid, hpre, hact, n1, n2;       // BYTE offsets of the slots inside a block's record; x is at 0
};
struct RecCtx {                  // one block's record as the phases see it
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned row0;               // first row of the workgroup's image
    unsigned qkv, ao, hpre, hact;
};
// A gfx950 hazard the compiler does not know (tools/microbench/store_hazard.hip, profiles/r05_store_hazard.md): a VALU write of the data
// registers of a buffer_store_dwordx4 WITH AN SGPR soffset, issued in the very next slot, reaches memory — dword 1 of lanes 12-15 of
// every row of 16 lanes carries the new value.  LLVM's createsVALUHazard covers only the form whose soffset is not a register (which
// needs two wait states, and gets them).  So the 16-byte record stores use THAT form: soffset 0, the uniform part added into the
// per-lane offset (one v_add_u32 per store).  (An asm statement "store; s_nop" is no way out: the hazard recogniser does not look
// into asm either, and a v_readlane feeding the asm's soffset in the slot before it is the next unhandled hazard.)
__device__ __forceinline__ void rec_store16(const RecCtx& rc, unsigned voff, unsigned soff, const u32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rc.rsrc, voff + soff, 0, 0);
}
__device__ __forceinline__ void rec_store16(const RecCtx& rc, unsigned voff, unsigned soff, const bf16x8& v) {
    rec_store16(rc, voff, soff, __builtin_bit_cast(u32x4, v));
}
__device__ __forceinline__ void rec_store16(const RecCtx& rc, unsigned voff, unsigned soff, float a, float b, float c, float d) {
    rec_store16(rc, voff, soff, u32x4{__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)});
}
// What a CU's store path costs is the number of 128-byte lines an instruction touches, not its bytes (tools/microbench/store_patterns.hip,
// profiles/r05_store_patterns.log: 8 rows x 128 B, 4 x 256 B or 1 KB contiguous per instruction 51 B/clk per CU; 16 rows x 64 B, or 16 rows
// x four 16-B pieces, 16 B/clk — and with one in-order wave per SIMD the store that finds the path busy holds the MFMA stream).  A lane of
// the fragment layouts owns 16 bytes of a row of which the four lanes (r16, 0..3) make 64: two pieces A, B that are neighbours in memory
// are therefore stored as TWO instructions of 8 rows x 128 B — the half-row swap of store_acc_to_x: rows r16 < 8 by the first (lanes
// r16 < 8 their own A, lanes r16 >= 8 the B of row r16 - 8), rows r16 >= 8 by the second.  GSTRIDE: bytes between the pieces of lanes g
// and g + 1; BSTRIDE: bytes from A to B.  `soff`: byte offset of (row tile's row 0, A's column of g = 0).
template <int GSTRIDE, int BSTRIDE>
__device__ __forceinline__ void rec_store_pair(const RecCtx& rc, int ln, unsigned pitch_b, unsigned soff, const u32x4& A, const u32x4& B) {
    const int r16 = ln & 15, g = ln >> 4;
    const bool lo_half = r16 < 8;
    const u32x4 got = swap_half_rows(lo_half ? B : A);
    const unsigned voff = (unsigned)(r16 & 7) * pitch_b + (unsigned)(g * GSTRIDE + (lo_half ? 0 : BSTRIDE));
    rec_store16(rc, voff, soff, lo_half ? A : got);
    rec_store16(rc, voff, soff + 8u * pitch_b, lo_half ? got : B);
}
// Record stores a wave issues BEHIND the LDS-DMA pieces of a ring triple: the ring-stage waits count them (vmcnt retires loads and stores
// in issue order on gfx9, so "all but the N youngest" lets exactly these stay in flight).  The waits are written in these constants and
// the loops that issue the stores are static_assert'ed against them — a change to a store helper cannot silently loosen a wait.
constexpr int kRecStoresPerPair = 2;                                    // rec_store_pair: two buffer_store_b128
constexpr int kRecRowTiles = 2;                                         // 16-row tiles per wave (32 rows)
template <int E> constexpr int kRecStoresFrag = kRecRowTiles * (E / 64) * kRecStoresPerPair;      // rec_store_frag: a LayerNorm output (24 at E = 384)
constexpr int kRecStoresQK = kRecRowTiles * 2 * kRecStoresPerPair;      // q or k epilogue of a head: 2 column pairs per row tile (8)
constexpr int kRecStoresV = 4 * kRecRowTiles * 4;                       // v epilogue of a head: 4 column groups x 2 row tiles x 4 b32 stores (32)
constexpr int kRecStoresAO = kRecRowTiles * kRecStoresPerPair;          // a head's attention output (4)
constexpr int kRecStoresMlpChunk = kRecRowTiles * 2 * kRecStoresPerPair;   // pre-activation + GELU of a 64-wide hidden chunk (8)
constexpr int kTriplePieces = 12;                                       // LDS-DMA pieces per wave of one triple of 16 KiB stages
static_assert(kRecStoresFrag<384> == 24 && kRecStoresQK == 8 && kRecStoresV + kRecStoresAO == 36 && kRecStoresMlpChunk == 8, "record-mode store counts");
__device__ __forceinline__ u32x4 rec_bits(const bf16x8& v) { return __builtin_bit_cast(u32x4, v); }
// bf16 operand fragments (lane (r16, g), row tile j, k-step ks: columns 32 ks + 8 g + [0, 8) of row 32 wid + 16 j + r16) -> [rows][E] bf16 at `slot`
template <int E>
__device__ __forceinline__ void rec_store_frag(const RecCtx& rc, unsigned slot, int wid, const bf16x8 (&afrag)[2][E / 32]) {
    const int ln = opaque_lane();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < E / 32; ks += 2)
            rec_store_pair<16, 64>(rc, ln, 2u * E, slot + ((rc.row0 + 32u * wid + 16u * j) * E + 32u * ks) * 2u, rec_bits(afrag[j][ks]), rec_bits(afrag[j][ks + 1]));
}
// the fp32 rows in the accumulators -> [rows][E] f32 at `slot` (store_acc_to_x's pieces through the descriptor)
template <int E>
__device__ __forceinline__ void rec_store_acc(const RecCtx& rc, unsigned slot, int wid, const f32x4 (&acc)[E / 16][2]) {
    const int ln = opaque_lane();
    const int rr = ln & 15, g = ln >> 4;
    const bool lo_half = rr < 8;
    const unsigned voff = (unsigned)((rr & 7) * E + 8 * g + (lo_half ? 0 : 4)) * 4u;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q32 = 0; q32 < E / 32; ++q32) {
            const f32x4 ta = acc[(q32 >> 2) * 8 + 2 * (q32 & 3)][j], tb = acc[(q32 >> 2) * 8 + 2 * (q32 & 3) + 1][j];
            const u32x4 pa = {__float_as_uint(ta[0]), __float_as_uint(ta[1]), __float_as_uint(ta[2]), __float_as_uint(ta[3])};
            const u32x4 pb = {__float_as_uint(tb[0]), __float_as_uint(tb[1]), __float_as_uint(tb[2]), __float_as_uint(tb[3])};
            const u32x4 got = swap_half_rows(lo_half ? pb : pa);
            const u32x4 first = lo_half ? pa : got, second = lo_half ? got : pb;
            const unsigned soff = slot + ((rc.row0 + 32u * wid + 16u * j) * E + 32u * q32) * 4u;
            rec_store16(rc, voff, soff, first);
            rec_store16(rc, voff, soff + 8u * E * 4u, second);
        }
}

// byte offset of lane (r16, g)'s first-k-step fragment inside a ring stage; the second k-step's is this ^ 64
__device__ __forceinline__ int stage_frag_off(int ln) { const int r16 = ln & 15, g_ = ln >> 4; return r16 * 128 + ((g_ ^ (r16 & 7)) << 4); }

// Per-lane constants of the weight stream (identical in both phases): the BYTE offset of this lane's DMA source relative to the
// wave-uniform stage origin, for the three (row order, row pitch) combinations the two phases use — computed once per kernel.
// Every DMA is
//     buffer_load_dwordx4 voffset, s[rsrc], soffset offen offset:imm lds
// — one descriptor over the whole bf16 weight pack, the stage origin (+ the piece's row delta) in an SGPR, ONE 32-bit VGPR for all
// four pieces of a stage.  (With 64-bit per-lane global pointers the compiler hoists four address pairs per matrix out of the
// loops, spills them, and every reload inside the loop carries an s_waitcnt vmcnt(0) that drains the LDS-DMA pipeline; with the
// offsets recomputed at every issue — the round's intermediate form — the address arithmetic was 5 % of the kernel.)
struct StreamLane {
    // A wave's four DMA pieces of a stage cover LDS rows 32 wid + 8 q + (lane >> 3), q = 0..3.  Under the pair permutation their
    // source rows are row(q) = row(0) + {0, 16, 4, 20}[q] for every lane, so ONE per-lane byte offset per (row order, row pitch)
    // combination is needed — three registers for the whole kernel — and the q-dependent part (a compile-time multiple of the
    // row pitch) goes into the scalar offset operand of the buffer load.  The four pieces share ONE M0 value: the instruction's
    // immediate offset q * 1024 moves the LDS destination (LDS address = M0 base + immediate + 16 lane) and is taken back out
    // of the memory address through the scalar offset.
    enum Kind { K64 = 0, K128 = 1 };     // 64 rows x two 64-k halves | 128 rows x 64 k
    unsigned v64, v128, v128w;           // K64 at pitch E | K128 at pitch E | K128 at pitch 4E
    __device__ __forceinline__ StreamLane(int lane, int wid, int E) {
        const int src_chunk = ((lane & 7) ^ (lane >> 3)) * 8;          // XOR swizzle on the source (LDS row & 7 == lane >> 3)
        const int rho = wid * 32 + (lane >> 3);                        // q = 0
        const int i = rho >> 4, r16 = rho & 15, i4 = i & 3;
        const int p64 = ((i4 >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i4 & 1) * 4 + (r16 & 3);
        const int p128 = (i >> 2) * 64 + ((i >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3);
        v64 = (unsigned)(p64 * E + (rho >> 6) * 64 + src_chunk) * 2u;
        v128 = (unsigned)(p128 * E + src_chunk) * 2u;
        v128w = (unsigned)(p128 * 4 * E + src_chunk) * 2u;
    }
    // `origin`: wave-uniform BYTE offset of (row 0, k 0) of the stage inside the buffer `rsrc` describes; pitch: row pitch in elements
    // (E, or 4E for KIND == K128 only)
    // q = -1: all four pieces of the wave's share of the stage; q = 0..3: that piece only
    template <int KIND, bool WIDE = false>
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc, unsigned origin, int pitch, unsigned char* dst, int q = -1) const {
        issue_v(rsrc, KIND == K64 ? v64 : (WIDE ? v128w : v128), origin, pitch, dst, q);
    }
    // the same with the per-lane offset given (a row pitch other than E / 4E: the patch-embedding weights)
    static __device__ __forceinline__ void issue_v(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned origin, int pitch, unsigned char* dst, int q = -1) {
        auto* l = (__attribute__((address_space(3))) void*)dst;
        const unsigned rp = 2u * (unsigned)pitch;
        if (q < 0 || q == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin, 0, 0);
        if (q < 0 || q == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin + 16u * rp - 1024u, 1024, 0);
        if (q < 0 || q == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin + 4u * rp - 2048u, 2048, 0);
        if (q < 0 || q == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, voff, origin + 20u * rp - 3072u, 3072, 0);
    }
    // per-lane byte offset of the K128 form at an arbitrary row pitch
    static __device__ __forceinline__ unsigned voff128(int lane, int wid, int pitch) {
        const int src_chunk = ((lane & 7) ^ (lane >> 3)) * 8;
        const int rho = wid * 32 + (lane >> 3);
        const int i = rho >> 4, r16 = rho & 15;
        const int p128 = (i >> 2) * 64 + ((i >> 1) & 1) * 32 + (r16 >> 2) * 8 + (i & 1) * 4 + (r16 & 3);
        return (unsigned)(p128 * pitch + src_chunk) * 2u;
    }
};

// ---- weight TRIPLES ----------------------------------------------------------------------------------------------------------
// Every GEMM of a block consumes its weights three 16 KiB stages at a time (a 64-wide q / k / v chunk, a 64-wide slice of proj, a
// hidden chunk's W1 rows or W2 columns are each 3 stages), so the ring is organised in GROUPS of three slots and a triple runs
// under ONE workgroup barrier: 96 MFMAs per wave between barriers instead of 32, and — with no barrier in the way — the fragment
// reads of a half stage are issued under the MFMAs of the half stage before it (two 8-register buffers, ping-pong).  The next
// triple's LDS-DMA pieces are issued at this triple's stage boundaries into a group whose last readers the triple's opening
// barrier has already retired.
constexpr int EB_GROUP_BYTES = 3 * 16384;
template <int N> __device__ __forceinline__ void eb_wait_vmcnt() { wait_vmcnt<N>(); }
// mma(k, half, i, w): the two MFMAs (row tiles j = 0, 1) that consume weight fragment i of k-half `half` of stage k.
// issue(k, q): the wave's LDS-DMA pieces due at stage k, one call per stage with q = -1 (all four pieces back to back) (the first
// after the first eight fragment reads of the triple have been issued); otherwise one call per stage with q = -1 (all four).
template <class Mma, class Issue>
__device__ __forceinline__ void run_triple(const unsigned char* grp, Mma&& mma, Issue&& issue) {
    const int ln = opaque_lane();
    const int fo0 = stage_frag_off(ln), fo1 = fo0 ^ 64;
    bf16x8 wa[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(grp + fo0 + i * 2048);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 6>([&](auto bc) {
        constexpr int b = decltype(bc)::value, k = b >> 1, half = b & 1, nb = b + 1;
        constexpr bool reads = b < 5;
        const unsigned char* src = grp + (nb >> 1) * 16384 + ((nb & 1) ? fo1 : fo0);
        static_for<0, 2>([&](auto sc) {
            constexpr int sub = decltype(sc)::value;
            if constexpr (half == 0 && sub == 0) {
                issue(k, -1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 4 * sub; i < 4 * sub + 4; ++i) {
                if constexpr (reads) {
                    if constexpr (half) wa[i] = *reinterpret_cast<const bf16x8*>(src + i * 2048);
                    else wb[i] = *reinterpret_cast<const bf16x8*>(src + i * 2048);
                }
                if constexpr (half) mma(k, half, i, wb[i]); else mma(k, half, i, wa[i]);
            }
            if constexpr (reads) {
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

// which stages of the NEXT triple go out at stage boundary k of this one: two, one, none (1 / 1 / 1 and 3 / 0 / 0 measured slower)
template <class F>
__device__ __forceinline__ void issue_split(int k, F&& one) {
    if (k == 0) { one(0); one(1); } else if (k == 1) one(2);
}

// ---- attention phase: acc += proj(attention(qkv(afrag)))  (bias of proj NOT added) ------------------------------------------
// LDS: ring groups 0-1 at `ring` (stage t of a head: group (t / 3) & 1, slot t % 3), K image at `kimg`, V^T image at `vimg`, qkv bias
// (3E floats) at `sbq`.  attn_prefetch must have been called (after a barrier that retired every earlier reader of groups 0-1);
// returns with no LDS-DMA in flight.  Every wave of the workgroup must call it (barriers inside).
template <int E>
__device__ __forceinline__ void attn_issue_stage(const StreamLane& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, unsigned wproj_off,
                                                 int wid, int h, int t, int q = -1) {
    constexpr int KS1 = E / 128;
    unsigned char* dst = ring + (((t / 3) & 1) * 3 + t % 3) * 16384 + wid * 4096;
    if (t < 3 * KS1) sl.template issue<StreamLane::K64>(wrsrc, (wqkv_off + (unsigned)(((t / KS1) * E + h * 64) * E + (t % KS1) * 128)) * 2u, E, dst, q);
    else sl.template issue<StreamLane::K128>(wrsrc, (wproj_off + (unsigned)((t - 3 * KS1) * 128 * E + h * 64)) * 2u, E, dst, q);
}
template <int E>
__device__ __forceinline__ void attn_prefetch(const StreamLane& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, int wid) {
    static_for<0, 3>([&](auto tc) { attn_issue_stage<E>(sl, ring, wrsrc, wqkv_off, 0u, wid, 0, decltype(tc)::value); });
}

template <int E, bool REC = false>
__device__ __forceinline__ void attn_phase(unsigned char* ring, unsigned char* kimg, unsigned char* vimg, const float* sbq,
                                           __amdgpu_buffer_rsrc_t wrsrc, unsigned wqkv_off, unsigned wproj_off, float scale,
                                           const StreamLane& sl, int wid, int rr, int g, const bf16x8 (&afrag)[2][E / 32],
                                           f32x4 (&acc2)[E / 16][2], const RecCtx& rc = RecCtx{}) {
    constexpr int H = E / 64, KS1 = E / 128, NG = E / 128;
    static_assert(E == 384 && KS1 == 3 && NG == 3, "written for E = 384: every chunk is one triple");
    const float sc2 = scale * 1.44269504088896340736f;

    for (int h = 0; h < H; ++h) {
        f32x4 acc1[4][2];
        bf16x8 qfrag[2][2], ofrag[2][2];
        static_for<0, 4>([&](auto uc) {
            constexpr int u = decltype(uc)::value;               // 0 q, 1 k, 2 v (operand roles swapped: V^T), 3 proj
            // this triple (issued during the previous one) has landed.  Record mode: the stores issued BEHIND its pieces — the previous
            // triple's epilogue (q: 8, k: 8, v and the attention output: 32 + 4), before head 0 the 24 of LayerNorm1's output — may stay in
            // flight (vmcnt retires loads and stores in issue order on gfx9); waiting for them too would expose a write's latency per triple
            if constexpr (!REC) eb_wait_vmcnt<0>();
            else if constexpr (u == 0) { if (h == 0) eb_wait_vmcnt<kRecStoresFrag<E>>(); else eb_wait_vmcnt<0>(); }
            else if constexpr (u == 3) eb_wait_vmcnt<kRecStoresV + kRecStoresAO>();
            else eb_wait_vmcnt<kRecStoresQK>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (u < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            auto issue = [&](int k, int q) {
                issue_split(k, [&](int sn) {
                    if constexpr (u < 3) attn_issue_stage<E>(sl, ring, wrsrc, wqkv_off, wproj_off, wid, h, 3 * (u + 1) + sn, q);
                    else if (h + 1 < H) attn_issue_stage<E>(sl, ring, wrsrc, wqkv_off, wproj_off, wid, h + 1, sn, q);
                });
            };
            const unsigned char* grp = ring + (u & 1) * EB_GROUP_BYTES;
            if constexpr (u < 2) {
                run_triple(grp, [&](int k, int half, int i, const bf16x8& w) {
                    acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, afrag[0][(2 * k + (i >> 2)) * 2 + half], acc1[i & 3][0], 0, 0, 0);
                    acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, afrag[1][(2 * k + (i >> 2)) * 2 + half], acc1[i & 3][1], 0, 0, 0);
                }, issue);
                const int ln = opaque_lane();
                const int rr = ln & 15, g = ln >> 4;              // (shadow the arguments: see opaque_lane)
                // K image row of this lane's token (32 wid + 16 j + r16): see encoder_attn_fused.h
                const int krow_j0 = 32 * wid + 16 * ((rr >> 2) & 1) + 4 * (rr >> 3) + (rr & 3);
                const float* bp0 = sbq + u * E + h * 64 + 8 * g;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        bf16x8 f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = static_cast<bf16_t>(acc1[2 * pr][j][r] + bp0[32 * pr + r]);
                            f[4 + r] = static_cast<bf16_t>(acc1[2 * pr + 1][j][r] + bp0[32 * pr + 4 + r]);
                        }
                        if constexpr (u == 0) qfrag[j][pr] = f;
                        else *reinterpret_cast<bf16x8*>(kimg + (krow_j0 + 8 * j) * AF_KROWB + 64 * pr + 16 * g) = f;
                        if constexpr (REC) {       // q | k with their biases, f32: columns u E + 64 h + 32 pr + 8 g + [0, 8) of row 32 wid + 16 j + r16
                            const u32x4 pa = {__float_as_uint(acc1[2 * pr][j][0] + bp0[32 * pr]), __float_as_uint(acc1[2 * pr][j][1] + bp0[32 * pr + 1]),
                                              __float_as_uint(acc1[2 * pr][j][2] + bp0[32 * pr + 2]), __float_as_uint(acc1[2 * pr][j][3] + bp0[32 * pr + 3])};
                            const u32x4 pb = {__float_as_uint(acc1[2 * pr + 1][j][0] + bp0[32 * pr + 4]), __float_as_uint(acc1[2 * pr + 1][j][1] + bp0[32 * pr + 5]),
                                              __float_as_uint(acc1[2 * pr + 1][j][2] + bp0[32 * pr + 6]), __float_as_uint(acc1[2 * pr + 1][j][3] + bp0[32 * pr + 7])};
                            rec_store_pair<32, 16>(rc, ln, 12u * E, rc.qkv + ((rc.row0 + 32u * wid + 16u * j) * (3u * E) + u * E + h * 64u + 32u * pr) * 4u, pa, pb);
                        }
                    }
            } else if constexpr (u == 2) {
                run_triple(grp, [&](int k, int half, int i, const bf16x8& w) {
                    acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[0][(2 * k + (i >> 2)) * 2 + half], w, acc1[i & 3][0], 0, 0, 0);
                    acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[1][(2 * k + (i >> 2)) * 2 + half], w, acc1[i & 3][1], 0, 0, 0);
                }, issue);
                const int ln = opaque_lane();
                const int rr = ln & 15, g = ln >> 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float bv = sbq[2 * E + h * 64 + ((i >> 1) & 1) * 32 + (rr >> 2) * 8 + (i & 1) * 4 + (rr & 3)];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float o4[4] = {acc1[i][j][0] + bv, acc1[i][j][1] + bv, acc1[i][j][2] + bv, acc1[i][j][3] + bv};
                        store4<bf16_t>(reinterpret_cast<bf16_t*>(vimg + (16 * i + rr) * AF_VROWB) + 32 * wid + 16 * j + 4 * g, o4);
                        if constexpr (REC) {       // v with its bias, f32: this lane's unit of tokens 32 wid + 16 j + 4 g + [0, 4) (the operand roles are swapped)
                            const unsigned voff = (unsigned)(4 * g * 3 * E + (rr >> 2) * 8 + (rr & 3)) * 4u;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o4[r]), rc.rsrc, voff,
                                    rc.qkv + ((rc.row0 + 32u * wid + 16u * j + r) * (3u * E) + 2u * E + h * 64u + ((i >> 1) & 1) * 32u + (i & 1) * 4u) * 4u, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                {
                // S^T = K Q^T and the soft-max, one 16-query row tile at a time (32 score registers live instead of 64; the K
                // fragments are read twice, 16 extra ds_read_b128 per head)
                bf16x8 pfrag[2][4];
                float inv[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 sc[8];
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt) sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int kt = 0; kt < 8; ++kt) {
                            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kimg + (16 * kt + rr) * AF_KROWB + 64 * ks + 16 * g);
                            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfrag[j][ks], sc[kt], 0, 0, 0);
                        }
                    float mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
                    mx = rows4_max(mx);
                    const float mc = mx * sc2;
                    float sum = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        bf16x8 f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p0 = __builtin_amdgcn_exp2f(sc[2 * ks][r] * sc2 - mc), p1 = __builtin_amdgcn_exp2f(sc[2 * ks + 1][r] * sc2 - mc);   // (v_exp_f32: a result below 2^-126 flushes to 0 instead of going through exp2f's denormal rescue)
                            sum += p0 + p1;
                            f[r] = static_cast<bf16_t>(p0);
                            f[4 + r] = static_cast<bf16_t>(p1);
                        }
                        pfrag[j][ks] = f;
                    }
                    sum = rows4_sum(sum);
                    inv[j] = 1.0f / sum;
                }
                f32x4 ov[4][2];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { ov[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[dt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vimg + (16 * dt + rr) * AF_VROWB + 64 * ks + 16 * g);
                        ov[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfrag[0][ks], ov[dt][0], 0, 0, 0);
                        ov[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfrag[1][ks], ov[dt][1], 0, 0, 0);
                    }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        bf16x8 f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f[r] = static_cast<bf16_t>(ov[2 * pr][j][r] * inv[j]);
                            f[4 + r] = static_cast<bf16_t>(ov[2 * pr + 1][j][r] * inv[j]);
                        }
                        ofrag[j][pr] = f;
                    }
                if constexpr (REC) {               // the attention output, bf16: columns 64 h + [0, 64) of rows 32 wid + 16 j + r16
                    const int ln = opaque_lane();
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        rec_store_pair<16, 64>(rc, ln, 2u * E, rc.ao + ((rc.row0 + 32u * wid + 16u * j) * E + h * 64u) * 2u, rec_bits(ofrag[j][0]), rec_bits(ofrag[j][1]));
                }
                }
            } else {
                run_triple(grp, [&](int k, int half, int i, const bf16x8& w) {
                    acc2[k * 8 + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, ofrag[0][half], acc2[k * 8 + i][0], 0, 0, 0);
                    acc2[k * 8 + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, ofrag[1][half], acc2[k * 8 + i][1], 0, 0, 0);
                }, issue);
            }
        });
    }
}

// ---- MLP phase: acc += fc2(gelu(fc1(afrag) + b1))  (bias of fc2 NOT added) --------------------------------------------------
// LDS: ring groups 0-2 at `ring` (144 KiB; triple n of the phase — fc1 of chunk n / 2 when n is even, fc2 when odd — lives in group
// n % 3), fc1 bias (4E floats) at `sb1`.  mlp_prefetch must have been called (after a barrier that retired every earlier reader of
// groups 0-1); returns with no LDS-DMA in flight.
template <int E>
__device__ __forceinline__ void mlp_issue_stage(const StreamLane& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                                int wid, int c, int t, int group, int q = -1) {
    constexpr int F = 4 * E, KS1 = E / 128;
    unsigned char* dst = ring + group * EB_GROUP_BYTES + (t % 3) * 16384 + wid * 4096;
    if (t < KS1) sl.template issue<StreamLane::K64>(wrsrc, (w1_off + (unsigned)(c * MLP_HC * E + t * 128)) * 2u, E, dst, q);
    else sl.template issue<StreamLane::K128, true>(wrsrc, (w2_off + (unsigned)((t - KS1) * 128 * F + c * MLP_HC)) * 2u, F, dst, q);
}
template <int E>
__device__ __forceinline__ void mlp_prefetch(const StreamLane& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off, int wid) {
    static_for<0, 6>([&](auto tc) { constexpr int t = decltype(tc)::value; mlp_issue_stage<E>(sl, ring, wrsrc, w1_off, w2_off, wid, 0, t, t / 3); });
}

template <int E, bool REC = false>
__device__ __forceinline__ void mlp_phase(unsigned char* ring, const float* sb1, __amdgpu_buffer_rsrc_t wrsrc, unsigned w1_off, unsigned w2_off,
                                          const StreamLane& sl, int wid, int rr, int g, const bf16x8 (&afrag)[2][E / 32],
                                          f32x4 (&acc2)[E / 16][2], const RecCtx& rc = RecCtx{}) {
    constexpr int F = 4 * E, KS1 = E / 128, KS2 = E / 128, NCH = F / MLP_HC;
    static_assert(KS1 == 3 && KS2 == 3, "every GEMM slice of a hidden chunk is one triple");
    int gcur = 0;                                                // group of the triple about to run
    for (int c = 0; c < NCH; ++c) {
        const bool more = c + 1 < NCH;
        f32x4 acc1[4][2];
        bf16x8 hfrag[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // ---- fc1 triple: in flight behind it is only this chunk's fc2 triple (12 pieces per wave) — and, in record mode before chunk 0,
        // the 24 stores of LayerNorm2's output issued behind the prefetch
        if (REC && c == 0) eb_wait_vmcnt<kTriplePieces + kRecStoresFrag<E>>(); else eb_wait_vmcnt<kTriplePieces>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        run_triple(ring + gcur * EB_GROUP_BYTES, [&](int k, int half, int i, const bf16x8& w) {
            acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, afrag[0][(2 * k + (i >> 2)) * 2 + half], acc1[i & 3][0], 0, 0, 0);
            acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, afrag[1][(2 * k + (i >> 2)) * 2 + half], acc1[i & 3][1], 0, 0, 0);
        }, [&](int k, int q) {
            if (more) mlp_issue_stage<E>(sl, ring, wrsrc, w1_off, w2_off, wid, c + 1, k, gcur == 0 ? 2 : gcur - 1, q);
        });
        {
            const int ln_ = opaque_lane();
            const int g = ln_ >> 4;                              // (shadows the argument: see opaque_lane)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x2 xv[8], yv[8];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float* bp = sb1 + c * MLP_HC + 32 * pr + 8 * g;
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        xv[4 * pr + h2] = f32x2{acc1[2 * pr][j][2 * h2] + bp[2 * h2], acc1[2 * pr][j][2 * h2 + 1] + bp[2 * h2 + 1]};
                        xv[4 * pr + 2 + h2] = f32x2{acc1[2 * pr + 1][j][2 * h2] + bp[4 + 2 * h2], acc1[2 * pr + 1][j][2 * h2 + 1] + bp[4 + 2 * h2 + 1]};
                    }
                }
                gelu_poly_16(xv, yv);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    bf16x8 f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f[2 * e] = static_cast<bf16_t>(yv[4 * pr + e][0]); f[2 * e + 1] = static_cast<bf16_t>(yv[4 * pr + e][1]); }
                    hfrag[j][pr] = f;
                }
                if constexpr (REC) {               // the pre-activation and its GELU, bf16: hidden units 64 c + [0, 64) of rows 32 wid + 16 j + r16
                    bf16x8 pre[2];
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pre[pr][2 * e] = static_cast<bf16_t>(xv[4 * pr + e][0]); pre[pr][2 * e + 1] = static_cast<bf16_t>(xv[4 * pr + e][1]); }
                    const unsigned soff = ((rc.row0 + 32u * wid + 16u * j) * F + c * MLP_HC) * 2u;
                    rec_store_pair<16, 64>(rc, ln_, 2u * F, rc.hpre + soff, rec_bits(pre[0]), rec_bits(pre[1]));
                    rec_store_pair<16, 64>(rc, ln_, 2u * F, rc.hact + soff, rec_bits(hfrag[j][0]), rec_bits(hfrag[j][1]));
                }
            }
        }
        gcur = gcur == 2 ? 0 : gcur + 1;
        // ---- fc2 triple: in flight behind it is only the next chunk's fc1 triple
        // (record mode: plus this chunk's 8 stores of the pre-activation and its GELU, issued behind the next chunk's fc1 pieces)
        if (more) eb_wait_vmcnt<kTriplePieces + (REC ? kRecStoresMlpChunk : 0)>(); else eb_wait_vmcnt<REC ? kRecStoresMlpChunk : 0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int gn2 = gcur == 0 ? 2 : gcur - 1;
        run_triple(ring + gcur * EB_GROUP_BYTES, [&](int k, int half, int i, const bf16x8& w) {
            acc2[k * 8 + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hfrag[0][half], acc2[k * 8 + i][0], 0, 0, 0);
            acc2[k * 8 + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hfrag[1][half], acc2[k * 8 + i][1], 0, 0, 0);
        }, [&](int k, int q) {
            if (more) mlp_issue_stage<E>(sl, ring, wrsrc, w1_off, w2_off, wid, c + 1, 3 + k, gn2, q);     // always 1/1/1: a whole chunk ahead
        });
        gcur = gcur == 2 ? 0 : gcur + 1;
    }
}

// ---- head: x = patches Wpe^T + (bias + pos_embed), the (4, 8) patch embedding of a 32 x 128 crop, straight into the accumulators ----
// Optional (images == nullptr: x is loaded instead).  A crop is 8 x 16 patches = the workgroup's 128 tokens; K = 3 * 4 * 8 = 96 with
// k = 32 c + 8 ky + kx (timm PatchEmbed's Conv2d weight flattened), so k-step s of the MFMA is channel s and a lane's eight k-slots
// (8 g + [0, 8)) are ONE run of eight pixels: row 4 gy + g, columns 8 gx .. 8 gx + 7 of channel s.  The 384 x 96 weight streams as
// six stages (three 128-row groups x two 64-k halves, the second half only 32 k wide: its upper k-step is never multiplied) = two
// triples.  posb: [128][E] f32 = pos_embed + bias (built once per plan); img_dtype: PARSEQ_BF16 / PARSEQ_F32 / PARSEQ_U8 as in
// include/parseq_hip.h (1 / 0 / 2) — u8 pixels get the reference transform ((v / 255 - 0.5) / 0.5, strhub/data/module.py:78-81)
// before the bf16 rounding, exactly as gemm.h's APatch loader does.
struct EncHeadParams {
    const void* images; int img_dtype;
    unsigned wpe;                  // element offset of patch_embed.proj.weight in the bf16 pack
    const float* posb;
};
constexpr int EB_IMG_F32 = 0, EB_IMG_BF16 = 1, EB_IMG_U8 = 2;      // = PARSEQ_F32 / PARSEQ_BF16 / PARSEQ_U8
// StreamLane::issue_v forms the scalar offset of a stage's third piece as origin + 4 * (row pitch in bytes) - 2048: with the head's
// 192-byte rows the weight must start at least 2048 - 768 bytes = 640 elements into the buffer the descriptor covers, or that offset
// wraps below zero and the piece reads as zeros.  (The E- and 4E-pitch streams of the blocks have 4 * pitch >= 3072 bytes: no condition.)
constexpr unsigned EB_HEAD_MIN_WPE = 640;

template <int E>
__device__ __forceinline__ void patch_head(const EncHeadParams& hp, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, int wid, int lane, int image,
                                           f32x4 (&acc)[E / 16][2]) {
    static_assert(E == 384, "three 128-row groups");
    constexpr int PK = 96, IH = 32, IW = 128;
    const int rr = lane & 15, g = lane >> 4;
    // the weight stream first (24 pieces per wave), then the table and the pixels
    const unsigned vpe = StreamLane::voff128(lane, wid, PK);
    static_for<0, 6>([&](auto sc) {
        constexpr int st = decltype(sc)::value, ng = st >> 1, kh = st & 1;
        StreamLane::issue_v(wrsrc, vpe, (hp.wpe + (unsigned)(ng * 128 * PK + kh * 64)) * 2u, PK, ring + st * 16384 + wid * 4096);
    });
    load_x_to_acc<E>(hp.posb, 0, 128, wid, rr, g, acc);
    bf16x8 pfrag[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int token = 32 * wid + 16 * j + rr, gy = token >> 4, gx = token & 15;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t e0 = (((size_t)image * 3 + c) * IH + gy * 4 + g) * IW + gx * 8;
            bf16x8 f;
            if (hp.img_dtype == EB_IMG_BF16) {
                f = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(hp.images) + e0);
            } else if (hp.img_dtype == EB_IMG_F32) {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(hp.images) + e0);
                const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(hp.images) + e0 + 4);
                f[0] = static_cast<bf16_t>(a.x); f[1] = static_cast<bf16_t>(a.y); f[2] = static_cast<bf16_t>(a.z); f[3] = static_cast<bf16_t>(a.w);
                f[4] = static_cast<bf16_t>(b.x); f[5] = static_cast<bf16_t>(b.y); f[6] = static_cast<bf16_t>(b.z); f[7] = static_cast<bf16_t>(b.w);
            } else {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(hp.images) + e0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned v = ((i < 4 ? u.x : u.y) >> (8 * (i & 3))) & 0xffu;
                    f[i] = static_cast<bf16_t>(((float)v / 255.0f - 0.5f) / 0.5f);
                }
            }
            pfrag[j][c] = f;
        }
    }
    static_for<0, 2>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (T == 0) eb_wait_vmcnt<12>(); else eb_wait_vmcnt<0>();       // (plain loads above were waited for by their uses)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        run_triple(ring + T * EB_GROUP_BYTES, [&](int k, int half, int i, const bf16x8& w) {
            const int st = 3 * T + k, ng = st >> 1, s32 = 2 * (st & 1) + half;       // constants after inlining
            if (s32 < 3) {
                acc[ng * 8 + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, pfrag[0][s32], acc[ng * 8 + i][0], 0, 0, 0);
                acc[ng * 8 + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, pfrag[1][s32], acc[ng * 8 + i][1], 0, 0, 0);
            }
        }, [](int, int) {});
    });
}

// ---- tail: K | V = LayerNorm_final(x) Wkv^T + bkv, head-split bf16, for the decoder's cross-attention ---------------------------
// Twelve 64-wide output chunks (six of K, six of V; a chunk = two 32-wide decoder heads), each one weight triple in the q / k
// chunk form of attn_phase, alternating between ring groups 0 and 1.  `sbkv`: the 2E biases in LDS.  The first triple must have
// been issued (kv_prefetch); returns with no LDS-DMA in flight.
template <int E>
__device__ __forceinline__ void kv_issue_stage(const StreamLane& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, int wid, int c, int t, int q = -1) {
    unsigned char* dst = ring + ((c & 1) * 3 + t) * 16384 + wid * 4096;
    sl.template issue<StreamLane::K64>(wrsrc, (wkv_off + (unsigned)(c * 64 * E + t * 128)) * 2u, E, dst, q);
}
template <int E>
__device__ __forceinline__ void kv_prefetch(const StreamLane& sl, unsigned char* ring, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, int wid) {
    static_for<0, 3>([&](auto tc) { kv_issue_stage<E>(sl, ring, wrsrc, wkv_off, wid, 0, decltype(tc)::value); });
}
template <int E>
__device__ __forceinline__ void kv_phase(unsigned char* ring, const float* sbkv, __amdgpu_buffer_rsrc_t wrsrc, unsigned wkv_off, const StreamLane& sl,
                                         int wid, int image, int heads, bf16_t* __restrict__ kmem, bf16_t* __restrict__ vmem,
                                         const bf16x8 (&afrag)[2][E / 32]) {
    constexpr int NC = 2 * E / 64;
    static_assert(E == 384, "written for E = 384: every 64-wide chunk is one triple");
    for (int c = 0; c < NC; ++c) {
        f32x4 acc1[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        eb_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        run_triple(ring + (c & 1) * EB_GROUP_BYTES, [&](int k, int half, int i, const bf16x8& w) {
            acc1[i & 3][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, afrag[0][(2 * k + (i >> 2)) * 2 + half], acc1[i & 3][0], 0, 0, 0);
            acc1[i & 3][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, afrag[1][(2 * k + (i >> 2)) * 2 + half], acc1[i & 3][1], 0, 0, 0);
        }, [&](int k, int q) {
            if (c + 1 < NC) issue_split(k, [&](int sn) { kv_issue_stage<E>(sl, ring, wrsrc, wkv_off, wid, c + 1, sn, q); });
        });
        // lane (r16, g), row tile j, pair pr: units 64 c + 32 pr + 8 g + [0, 8) of token 32 wid + 16 j + r16 — one 16-byte piece of
        // the (token, head 2 c' + pr) row of K (c < NC / 2) or V
        const int ln = opaque_lane();
        const int rr = ln & 15, g = ln >> 4;
        bf16_t* dst = c < NC / 2 ? kmem : vmem;
        const int cc = c < NC / 2 ? c : c - NC / 2;
        const float* bp0 = sbkv + c * 64 + 8 * g;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                bf16x8 f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f[r] = static_cast<bf16_t>(acc1[2 * pr][j][r] + bp0[32 * pr + r]);
                    f[4 + r] = static_cast<bf16_t>(acc1[2 * pr + 1][j][r] + bp0[32 * pr + 4 + r]);
                }
                const int token = 32 * wid + 16 * j + rr;
                *reinterpret_cast<bf16x8*>(dst + (((size_t)image * heads + 2 * cc + pr) * 128 + token) * 32 + 8 * g) = f;
            }
    }
}

// Copy `n` floats from global memory into LDS (all 256 threads; plain loads: call only while no LDS-DMA is in flight).
__device__ __forceinline__ void params_to_lds(float* dst, const float* __restrict__ src, int n, int tid) {
    for (int i = tid; i < n; i += 256) dst[i] = src[i];
}

// REC (record mode, see EncRecordParams): x is read from block 0's record slot (where the patch embedding left it), every block's
// tensors are written on the way, the final rows go to `x`; no head, no tail; M must be a multiple of 128.
template <int E, bool REC = false>
__global__ __launch_bounds__(256, 1)
void enc_blocks_kernel(float* __restrict__ x, const bf16_t* __restrict__ wbase, unsigned wbytes, const float* __restrict__ pbase,
                       const EncBlockParams* __restrict__ blocks, int depth, float eps, int M, const EncTailParams tail, const EncHeadParams head,
                       const EncRecordParams rec) {
    constexpr int F = 4 * E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                                  // groups 0-1 (attention) / 0-2 (MLP)
    unsigned char* kimg = smem + 2 * EB_GROUP_BYTES;             // overlays ring group 2
    unsigned char* vimg = kimg + 128 * AF_KROWB;
    float* sp = reinterpret_cast<float*>(smem + EB_RING_BYTES);  // phase parameters, <= 7E floats
    static_assert(128 * AF_KROWB + 64 * AF_VROWB <= EB_GROUP_BYTES, "overlay region");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const StreamLane sl(lane, wid, E);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wbase), 0, wbytes, 0x00020000);

    f32x4 acc[E / 16][2];
    bf16x8 afrag[2][E / 32];
    if constexpr (REC) load_x_to_acc<E>(rec.base, m0, M, wid, rr, g, acc);
    else if (head.images) patch_head<E>(head, ring, wrsrc, wid, lane, blockIdx.x, acc);
    else load_x_to_acc<E>(x, m0, M, wid, rr, g, acc);

    for (int l = 0; l < depth; ++l) {
        const EncBlockParams* bp = blocks + l;
        RecCtx rc{};
        if constexpr (REC) {
            rc.rsrc = __builtin_amdgcn_make_buffer_rsrc(rec.base + (size_t)l * rec.layer_stride, 0, rec.layer_bytes, 0x00020000);
            rc.row0 = (unsigned)m0; rc.qkv = rec.qkv; rc.ao = rec.ao; rc.hpre = rec.hpre; rc.hact = rec.hact;
            if (l > 0) rec_store_acc<E>(rc, 0u, wid, acc);         // this block's input rows (block 0's are where they were read from)
        }
        // ---- attention branch: parameters bqkv (3E) | bproj (E) | ln1 gamma (E) | ln1 beta (E)
        __syncthreads();                                         // everyone is done with the previous phase's parameters and ring
        attn_prefetch<E>(sl, ring, wrsrc, bp->wqkv, wid);            // head 0's q triple lands behind the parameter copies and LayerNorm
        params_to_lds(sp, pbase + bp->bqkv, 3 * E, tid);
        params_to_lds(sp + 3 * E, pbase + bp->bproj, E, tid);
        params_to_lds(sp + 4 * E, pbase + bp->ln1_w, E, tid);
        params_to_lds(sp + 5 * E, pbase + bp->ln1_b, E, tid);
        __syncthreads();
        ln_acc_to_frag<E>(acc, sp + 4 * E, sp + 5 * E, eps, g, afrag);
        if constexpr (REC) rec_store_frag<E>(rc, rec.n1, wid, afrag);
        attn_phase<E, REC>(ring, kimg, vimg, sp, wrsrc, bp->wqkv, bp->wproj, 0.125f, sl, wid, rr, g, afrag, acc, rc);
        add_bias_to_acc<E>(sp + 3 * E, g, acc);
        if constexpr (REC) rec_store_acc<E>(rc, rec.x_mid, wid, acc);
        // ---- MLP branch: parameters b1 (4E) | b2 (E) | ln2 gamma (E) | ln2 beta (E)
        __syncthreads();
        mlp_prefetch<E>(sl, ring, wrsrc, bp->w1, bp->w2, wid);       // chunk 0's two triples
        params_to_lds(sp, pbase + bp->b1, F, tid);
        params_to_lds(sp + F, pbase + bp->b2, E, tid);
        params_to_lds(sp + F + E, pbase + bp->ln2_w, E, tid);
        params_to_lds(sp + F + 2 * E, pbase + bp->ln2_b, E, tid);
        __syncthreads();
        ln_acc_to_frag<E>(acc, sp + F + E, sp + F + 2 * E, eps, g, afrag);
        if constexpr (REC) rec_store_frag<E>(rc, rec.n2, wid, afrag);
        mlp_phase<E, REC>(ring, sp, wrsrc, bp->w1, bp->w2, sl, wid, rr, g, afrag, acc, rc);
        add_bias_to_acc<E>(sp + F, g, acc);
    }
    if (REC || tail.kmem == nullptr) {
        store_acc_to_x<E>(x, m0, M, wid, rr, g, acc);
        return;
    }
    // ---- tail: parameters bkv (2E) | final norm gamma (E) | beta (E)
    __syncthreads();
    kv_prefetch<E>(sl, ring, wrsrc, tail.wkv, wid);
    params_to_lds(sp, pbase + tail.bkv, 2 * E, tid);
    params_to_lds(sp + 2 * E, pbase + tail.norm_w, E, tid);
    params_to_lds(sp + 3 * E, pbase + tail.norm_b, E, tid);
    __syncthreads();
    ln_acc_to_frag<E>(acc, sp + 2 * E, sp + 3 * E, eps, g, afrag);
    kv_phase<E>(ring, sp, wrsrc, tail.wkv, sl, wid, blockIdx.x, tail.heads, tail.kmem, tail.vmem, afrag);
}

// The two branches as stand-alone launches built from the SAME phase functions as enc_blocks_kernel (per-kernel parity tests and
// the bit-exactness test of the one-launch encoder against a chain of these cover the shared code).
template <int E>
__global__ __launch_bounds__(256, 1)
void attn_branch_kernel(float* __restrict__ x, const bf16_t* __restrict__ wbase, unsigned wbytes, unsigned wqkv_off, unsigned wproj_off,
                        const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ bqkv,
                        const float* __restrict__ bproj, float eps, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    unsigned char* kimg = smem + 2 * EB_GROUP_BYTES;
    unsigned char* vimg = kimg + 128 * AF_KROWB;
    float* sp = reinterpret_cast<float*>(smem + EB_RING_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const StreamLane sl(lane, wid, E);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wbase), 0, wbytes, 0x00020000);
    f32x4 acc[E / 16][2];
    bf16x8 afrag[2][E / 32];
    load_x_to_acc<E>(x, m0, M, wid, rr, g, acc);
    attn_prefetch<E>(sl, ring, wrsrc, wqkv_off, wid);
    params_to_lds(sp, bqkv, 3 * E, tid);
    params_to_lds(sp + 3 * E, bproj, E, tid);
    params_to_lds(sp + 4 * E, gamma, E, tid);
    params_to_lds(sp + 5 * E, beta, E, tid);
    __syncthreads();
    ln_acc_to_frag<E>(acc, sp + 4 * E, sp + 5 * E, eps, g, afrag);
    attn_phase<E>(ring, kimg, vimg, sp, wrsrc, wqkv_off, wproj_off, 0.125f, sl, wid, rr, g, afrag, acc);
    add_bias_to_acc<E>(sp + 3 * E, g, acc);
    store_acc_to_x<E>(x, m0, M, wid, rr, g, acc);
}

template <int E>
__global__ __launch_bounds__(256, 1)
void mlp_branch_kernel(float* __restrict__ x, const bf16_t* __restrict__ wbase, unsigned wbytes, unsigned w1_off, unsigned w2_off,
                       const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ b1,
                       const float* __restrict__ b2, float eps, int M) {
    constexpr int F = 4 * E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* sp = reinterpret_cast<float*>(smem + EB_RING_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 128;
    const StreamLane sl(lane, wid, E);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wbase), 0, wbytes, 0x00020000);
    f32x4 acc[E / 16][2];
    bf16x8 afrag[2][E / 32];
    load_x_to_acc<E>(x, m0, M, wid, rr, g, acc);
    mlp_prefetch<E>(sl, ring, wrsrc, w1_off, w2_off, wid);
    params_to_lds(sp, b1, F, tid);
    params_to_lds(sp + F, b2, E, tid);
    params_to_lds(sp + F + E, gamma, E, tid);
    params_to_lds(sp + F + 2 * E, beta, E, tid);
    __syncthreads();
    ln_acc_to_frag<E>(acc, sp + F + E, sp + F + 2 * E, eps, g, afrag);
    mlp_phase<E>(ring, sp, wrsrc, w1_off, w2_off, sl, wid, rr, g, afrag, acc);
    add_bias_to_acc<E>(sp + F, g, acc);
    store_acc_to_x<E>(x, m0, M, wid, rr, g, acc);
}

// Two bf16 matrices addressed through one descriptor: base = the lower of the two addresses, element offsets from it.
struct WPair { const bf16_t* base; size_t bytes; unsigned off_a, off_b; };
inline bool make_wpair(const bf16_t* a, size_t a_elems, const bf16_t* b, size_t b_elems, WPair* out) {
    const uintptr_t pa = reinterpret_cast<uintptr_t>(a), pb = reinterpret_cast<uintptr_t>(b);
    const uintptr_t lo = pa < pb ? pa : pb, hi = (pa + a_elems * 2 > pb + b_elems * 2) ? pa + a_elems * 2 : pb + b_elems * 2;
    if (hi - lo >= ((uintptr_t)1 << 32) || ((pa | pb) & 1)) return false;
    out->base = reinterpret_cast<const bf16_t*>(lo); out->bytes = (size_t)(hi - lo);
    out->off_a = (unsigned)((pa - lo) / 2); out->off_b = (unsigned)((pb - lo) / 2);
    return true;
}

template <int E>
hipError_t launch_attn_branch(hipStream_t s, float* x, const float* gamma, const float* beta, float eps, const bf16_t* Wqkv, const float* bqkv,
                                     const bf16_t* Wproj, const float* bproj, int M) {
    WPair wp;
    if (!make_wpair(Wqkv, (size_t)3 * E * E, Wproj, (size_t)E * E, &wp)) return hipErrorInvalidValue;
    constexpr size_t lds = enc_blocks_lds<E>();
    auto kern = attn_branch_kernel<E>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((M + 127) / 128), dim3(256), lds, s, x, wp.base, (unsigned)wp.bytes, wp.off_a, wp.off_b, gamma, beta, bqkv, bproj, eps, M);
    return hipGetLastError();
}

template <int E>
hipError_t launch_mlp_branch(hipStream_t s, float* x, const float* gamma, const float* beta, float eps, const bf16_t* W1, const float* b1,
                                    const bf16_t* W2, const float* b2, int M) {
    WPair wp;
    if (!make_wpair(W1, (size_t)4 * E * E, W2, (size_t)4 * E * E, &wp)) return hipErrorInvalidValue;
    constexpr size_t lds = enc_blocks_lds<E>();
    auto kern = mlp_branch_kernel<E>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((M + 127) / 128), dim3(256), lds, s, x, wp.base, (unsigned)wp.bytes, wp.off_a, wp.off_b, gamma, beta, b1, b2, eps, M);
    return hipGetLastError();
}

template <int E>
hipError_t launch_enc_blocks(hipStream_t s, float* x, const bf16_t* wbase, size_t wbytes, const float* pbase, const EncBlockParams* blocks,
                                    int depth, float eps, int M, const EncTailParams& tail = EncTailParams{0, 0, 0, 0, nullptr, nullptr, 0},
                                    const EncHeadParams& head = EncHeadParams{nullptr, 0, 0, nullptr}) {
    constexpr size_t lds = enc_blocks_lds<E>();
    if (wbytes >= ((size_t)1 << 32)) return hipErrorInvalidValue;       // one 32-bit buffer descriptor covers the weight pack
    auto kern = enc_blocks_kernel<E>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((M + 127) / 128), dim3(256), lds, s, x, wbase, (unsigned)wbytes, pbase, blocks, depth, eps, M, tail, head, EncRecordParams{});
    return hipGetLastError();
}
// The record-mode walk (training forward): rows rec.base[0 .. M) in, rows x_last out, every block's record written.
template <int E>
hipError_t launch_enc_blocks_record(hipStream_t s, float* x_last, const bf16_t* wbase, size_t wbytes, const float* pbase, const EncBlockParams* blocks,
                                    int depth, float eps, int M, const EncRecordParams& rec) {
    constexpr size_t lds = enc_blocks_lds<E>();
    if (wbytes >= ((size_t)1 << 32) || M % 128 || !rec.base) return hipErrorInvalidValue;
    auto kern = enc_blocks_kernel<E, true>;
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(M / 128), dim3(256), lds, s, x_last, wbase, (unsigned)wbytes, pbase, blocks, depth, eps, M,
                       EncTailParams{0, 0, 0, 0, nullptr, nullptr, 0}, EncHeadParams{nullptr, 0, 0, nullptr}, rec);
    return hipGetLastError();
}

// The three kernels of this header are compiled in their own translation unit (kern_enc_blocks.hip defines PQ_INSTANTIATE_ENC_BLOCKS
// and instantiates the launchers for E = 384); every other unit only calls them.
#ifdef PQ_INSTANTIATE_ENC_BLOCKS
#define PQ_ENC_BLOCKS_EXTERN
#else
#define PQ_ENC_BLOCKS_EXTERN extern
#endif
PQ_ENC_BLOCKS_EXTERN template hipError_t launch_attn_branch<384>(hipStream_t, float*, const float*, const float*, float, const bf16_t*, const float*, const bf16_t*,
                                                                   const float*, int);
PQ_ENC_BLOCKS_EXTERN template hipError_t launch_mlp_branch<384>(hipStream_t, float*, const float*, const float*, float, const bf16_t*, const float*, const bf16_t*,
                                                                  const float*, int);
PQ_ENC_BLOCKS_EXTERN template hipError_t launch_enc_blocks<384>(hipStream_t, float*, const bf16_t*, size_t, const float*, const EncBlockParams*, int, float, int,
                                                                  const EncTailParams&, const EncHeadParams&);
PQ_ENC_BLOCKS_EXTERN template hipError_t launch_enc_blocks_record<384>(hipStream_t, float*, const bf16_t*, size_t, const float*, const EncBlockParams*, int, float, int,
                                                                         const EncRecordParams&);
#undef PQ_ENC_BLOCKS_EXTERN

}  // namespace pq
