// Row-wise kernels: LayerNorm as a wavefront reduction, token-embedding + LayerNorm for the decoder content table,
// greedy argmax / EOS bookkeeping, refinement token prep.  All wave64; one wave per row, 4 rows per workgroup.
#pragma once
#include "common.h"

namespace pq {

// LayerNorm over the last dim E (E = 192 * VEC, VEC in {1, 2, 4}).  x fp32 [M, E] -> out TO [M, E] and optionally a
// second fp32 copy (the encoder's final norm is both the API's `memory` output and the decoder K/V GEMM operand).
// torch.nn.LayerNorm semantics: biased variance, y = (x - mean) / sqrt(var + eps) * w + b  (ViT eps 1e-6, decoder 1e-5).
// out[r][c] = a[r][c] + v[c]   (pos_embed + patch-embed bias: the table the one-launch encoder's head starts its accumulators from)
static __global__ __launch_bounds__(256)
void add_rowvec_kernel(const float* __restrict__ a, const float* __restrict__ v, float* __restrict__ out, int rows, int cols) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)rows * cols) out[i] = a[i] + v[i % cols];
}

// LayerNorm statistics only: stats[2 m] = mean, stats[2 m + 1] = 1 / sqrt(var + eps) of row m (wave per row, the same two passes
// as layernorm_kernel).  Feeds gemm.h's ALayerNormStats loader: the rows are read once more by the GEMM, nothing is written back.
template <int E>
__global__ __launch_bounds__(256)
void ln_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, int M, float eps) {
    constexpr int VEC = E / 192;
    static_assert(E % 192 == 0 && (VEC == 1 || VEC == 2 || VEC == 4), "unsupported embed dim");
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * E;
    float v[3][VEC];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int e0 = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { v[it][j] = xr[e0 + j]; s += v[it][j]; }
    }
    const float mean = wave_sum(s) * (1.0f / E);
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float d = v[it][j] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / E) + eps);
    if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
}

// LayerNorm whose output is written as block-planar hi | lo bf16 pairs (bf16x3 mode: the A operand of a PAIRS GEMM, gemm.h): the 32
// elements [32 b, 32 b + 32) of a row occupy bytes [128 b, 128 b + 64) (hi) and [128 b + 64, 128 b + 128) (lo) of the row.
template <int E>
__global__ __launch_bounds__(256)
void layernorm_split_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                            unsigned char* __restrict__ out, int M, float eps) {
    constexpr int VEC = E / 192;
    static_assert(E % 192 == 0 && (VEC == 2 || VEC == 4), "unsupported embed dim");
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * E;
    float v[3][VEC];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int e0 = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { v[it][j] = xr[e0 + j]; s += v[it][j]; }
    }
    const float mean = wave_sum(s) * (1.0f / E);
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float d = v[it][j] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / E) + eps);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int e0 = (it * 64 + lane) * VEC;
        bf16_t hi[VEC], lo[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float y = (v[it][j] - mean) * rstd * w[e0 + j] + b[e0 + j];
            hi[j] = static_cast<bf16_t>(y);
            lo[j] = static_cast<bf16_t>(y - static_cast<float>(hi[j]));
        }
        unsigned char* d = out + ((size_t)row * E + (e0 & ~31)) * 4 + (e0 & 31) * 2;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            *reinterpret_cast<bf16_t*>(d + 2 * j) = hi[j];
            *reinterpret_cast<bf16_t*>(d + 64 + 2 * j) = lo[j];
        }
    }
}

template <typename TO, int E>
__global__ __launch_bounds__(256)
void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                      TO* __restrict__ out, float* __restrict__ out_f32, int M, float eps) {
    constexpr int VEC = E / 192;
    static_assert(E % 192 == 0 && (VEC == 1 || VEC == 2 || VEC == 4), "unsupported embed dim");
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * E;
    float v[3][VEC];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int e0 = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { v[it][j] = xr[e0 + j]; s += v[it][j]; }
    }
    const float mean = wave_sum(s) * (1.0f / E);
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float d = v[it][j] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / E) + eps);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int e0 = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float y = (v[it][j] - mean) * rstd * w[e0 + j] + b[e0 + j];
            out[(size_t)row * E + e0 + j] = from_f32<TO>(y);
            if (out_f32) out_f32[(size_t)row * E + e0 + j] = y;
        }
    }
}

// Decoder content rows for every (position j, token id): c = (j ? pos_queries[j-1] : 0) + sqrt(E) * emb[tok], then
// norm_c.  (model.py:97-98 + modules.py:175-176 + modules.py:91.)  Row index = j * ntok + tok.
template <typename TO, int E>
__global__ __launch_bounds__(256)
void content_ln_kernel(const float* __restrict__ emb, const float* __restrict__ posq, const float* __restrict__ w,
                       const float* __restrict__ b, TO* __restrict__ out, int npos, int ntok, float eps) {
    constexpr int VEC = E / 192;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= npos * ntok) return;
    const int j = row / ntok, tok = row - j * ntok;
    const float sq = sqrtf((float)E);
    float v[3][VEC];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int e0 = (it * 64 + lane) * VEC;
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) {
            float c = sq * emb[(size_t)tok * E + e0 + jj];
            if (j > 0) c = posq[(size_t)(j - 1) * E + e0 + jj] + c;
            v[it][jj] = c; s += c;
        }
    }
    const float mean = wave_sum(s) * (1.0f / E);
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) { const float d = v[it][jj] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / E) + eps);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int e0 = (it * 64 + lane) * VEC;
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj)
            out[(size_t)row * E + e0 + jj] = from_f32<TO>((v[it][jj] - mean) * rstd * w[e0 + jj] + b[e0 + jj]);
    }
}

// Greedy pick after AR step `step` (model.py:142-145): tok[b][step + 1] = argmax_c logits[b][step][c] (first max on
// ties, as torch.argmax), and the batch-level early-exit test kept on the device: the first step at which every row
// holds an EOS sets *ar_len = step + 1 (the number of logit positions the reference would have produced).
static __global__ __launch_bounds__(256)
void ar_argmax_kernel(const float* __restrict__ logits, int L, int C, int* __restrict__ tok, int ldt, int step,
                      int B, int eos_id, unsigned char* __restrict__ eos_seen, int* __restrict__ eos_rows,
                      int* __restrict__ ar_len, int record) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* row = logits + ((size_t)b * L + step) * C;
    float best = -INFINITY; int bi = ARGMAX_NONE;
    for (int c = lane; c < C; c += 64) {
        const float v = row[c];
        if (argmax_take(v, c, best, bi)) { best = v; bi = c; }
    }
    wave_argmax(best, bi);
    bi = argmax_final(bi, C);
    if (lane == 0) {
        tok[(size_t)b * ldt + step + 1] = bi;
        if (record && bi == eos_id && !eos_seen[b]) {
            eos_seen[b] = 1;
            const int done = atomicAdd(eos_rows, 1) + 1;
            if (done == B) *ar_len = step + 1;
        }
    }
}

// Caller-supplied context tokens (parseq_decode_logits / parseq_decode_hidden) index the decoder tables: ids outside
// [0, ntok) are clamped into range instead of reading out of bounds (the reference raises on the host or trips a device
// assert; this library cannot raise without a synchronisation, so it degrades to a valid id).
static __global__ void clamp_tokens_kernel(int* __restrict__ tok, int ldt, int B, int Lk, int ntok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Lk) return;
    int* p = tok + (size_t)(i / Lk) * ldt + (i % Lk);
    const int v = *p;
    *p = v < 0 ? 0 : (v >= ntok ? ntok - 1 : v);
}

// Refinement context (model.py:161-163): tok[b] = [bos, argmax(logits[b, :L-1])], and the key-padding mask
// kpm[b][j] = (an EOS occurs at a position <= j).  One wave per image; lane p handles logits position p (L <= 64).
static __global__ __launch_bounds__(256)
void refine_prep_kernel(const float* __restrict__ logits, int L, int C, int* __restrict__ tok, int ldt,
                        unsigned char* __restrict__ kpm, int ldk, int B, int bos_id, int eos_id, int from_logits) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int t = bos_id;
    if (lane >= 1 && lane < L) {
        if (from_logits) {
            const float* row = logits + ((size_t)b * L + (lane - 1)) * C;
            float best = row[0]; int bi = 0;
            for (int c = 1; c < C; ++c) { const float v = row[c]; if (argmax_take(v, c, best, bi)) { best = v; bi = c; } }
            t = bi;
        } else {
            t = tok[(size_t)b * ldt + lane];      // first iteration after a full AR loop: the AR picks are the argmaxes
        }
    }
    const unsigned long long eos_lanes = __ballot(lane < L && t == eos_id);
    if (lane < L) {
        tok[(size_t)b * ldt + lane] = t;
        const unsigned long long upto = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        kpm[(size_t)b * ldk + lane] = (eos_lanes & upto) ? 1 : 0;
    }
}

// counters: ncounters ints, pairs (rows that hold an EOS, step count the reference would have returned) per AR chain
static __global__ void ar_init_kernel(int* __restrict__ tok, int ldt, int B, int bos_id, int pad_id, unsigned char* __restrict__ eos_seen,
                               int* __restrict__ counters, int ncounters, int num_steps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * ldt) tok[i] = (i % ldt == 0) ? bos_id : pad_id;
    if (i < B) eos_seen[i] = 0;
    if (i < ncounters) counters[i] = (i & 1) ? num_steps : 0;
}

// Device-side numeric half of `Tokenizer.decode(logits.softmax(-1))` (strhub/models/base.py:132-135,
// strhub/data/utils.py:79-99 greedy max per position, :120-129 cut at the first EOS keeping the EOS probability).
// One wave per image, lane l ends up owning position l (L <= 64):
//   ids[b][l]   = argmax_c logits[b][l][c]  (first maximum; soft-max is monotone, so this is the arg-max of the probabilities)
//   probs[b][l] = max_c softmax(logits[b][l])[c] = 1 / sum_c exp(logit_c - max)
//   lengths[b]  = index of the first EOS, or L            (number of characters)
//   conf[b]     = prod_{l < min(lengths[b] + 1, L)} probs[b][l]   (base.py:137 `prob.prod()`)
static __global__ __launch_bounds__(256)
void postprocess_kernel(const float* __restrict__ logits, int B, int L, int C, int eos_id, int* __restrict__ ids,
                        int* __restrict__ lengths, float* __restrict__ probs, float* __restrict__ conf) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float my_p = 1.0f; int my_id = -1;
    for (int l = 0; l < L; ++l) {
        const float* row = logits + ((size_t)b * L + l) * C;
        float best = -INFINITY; int bi = ARGMAX_NONE;
        for (int c = lane; c < C; c += 64) {
            const float v = row[c];
            if (argmax_take(v, c, best, bi)) { best = v; bi = c; }
        }
        wave_argmax(best, bi);
        bi = argmax_final(bi, C);
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += expf(row[c] - best);
        sum = wave_sum(sum);
        if (lane == l) { my_p = 1.0f / sum; my_id = bi; }
    }
    const unsigned long long eos_lanes = __ballot(lane < L && my_id == eos_id);
    const int eos_idx = eos_lanes ? __builtin_ctzll(eos_lanes) : L;
    const int n_probs = min(eos_idx + 1, L);
    float prod = lane < n_probs ? my_p : 1.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) prod *= __shfl_xor(prod, o, 64);
    if (lane < L) {
        ids[(size_t)b * L + lane] = my_id;
        if (probs) probs[(size_t)b * L + lane] = my_p;
    }
    if (lane == 0) {
        lengths[b] = eos_idx;
        if (conf) conf[b] = prod;
    }
}

// Validation loss (strhub/models/base.py:194-201, CrossEntropySystem.forward_logits_loss):
//   F.cross_entropy(logits.flatten(end_dim=1), targets.flatten(), ignore_index=pad_id)  — mean over the non-ignored rows.
// Pass 1: one wave per row, row_loss[r] = logsumexp(logits[r]) - logits[r][target[r]] (0 for ignored rows).
// Pass 2: one workgroup sums the rows in a fixed order (deterministic) and writes the mean and the count.
static __global__ __launch_bounds__(256)
void ce_rows_kernel(const float* __restrict__ logits, const int* __restrict__ targets, int rows, int C, int ignore_index,
                    float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int tgt = targets[r];
    if (tgt == ignore_index) { if (lane == 0) row_loss[r] = 0.f; return; }
    const float* row = logits + (size_t)r * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(row[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) row_loss[r] = (mx + logf(sum)) - row[tgt];
}

static __global__ __launch_bounds__(256)
void ce_reduce_kernel(const float* __restrict__ row_loss, const int* __restrict__ targets, int rows, int ignore_index,
                      float* __restrict__ loss_out, int* __restrict__ numel_out) {
    __shared__ float ssum[256];
    __shared__ int scnt[256];
    float s = 0.f; int n = 0;
    for (int r = threadIdx.x; r < rows; r += 256) { s += row_loss[r]; n += targets[r] != ignore_index; }
    ssum[threadIdx.x] = s; scnt[threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *loss_out = ssum[0] / (float)scnt[0]; *numel_out = scnt[0]; }   // 0 / 0 = NaN, as torch
}

}  // namespace pq
