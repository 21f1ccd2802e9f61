// libparseq_hip.so — the training step.
#include "lib_internal.h"

#include "train_ops.h"

// -------------------------------------------------------------------------------------------------------------------
// training step, decoder side: loss of system.py:168-199 and its gradients, fp32
// -------------------------------------------------------------------------------------------------------------------
// Scratch shared by the split-K partials of the MFMA GEMM and the partial column sums; part of the caller's workspace.
constexpr size_t TRAIN_SCRATCH_FLOATS = (size_t)16 << 20;
// LayerNorm backward parks rows / LNB_ROWS partial rows of 2E floats at the END of the scratch and folds them through 128 E floats at its start,
// so a layout whose LayerNorms see more than ~87 k rows (E = 384) takes a larger scratch instead of failing.
static size_t train_scratch_floats(size_t ln_rows, size_t E) {
    return std::max(TRAIN_SCRATCH_FLOATS, ((ln_rows + LNB_ROWS - 1) / LNB_ROWS * 2 * E + 128 * E + 63) / 64 * 64);
}
struct TrainCtx {
    hipStream_t s;
    float* scratch;      // scratch_floats floats (train_scratch_floats)
    bool bf16_ops = false;      // GEMM operands rounded to bf16 (parseq_model_set_train_precision), fp32 accumulate and everything else
    size_t scratch_floats = TRAIN_SCRATCH_FLOATS;      // what of `scratch` the split-K partials / column sums may use (lin_bwd carves its padded copies off the end)
};

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// split-K: how many workgroups a product with few output tiles is cut into along the contraction (PARSEQ_TRAIN_SPLIT_TARGET overrides, for A/B)
static int split_target() {
    static const int t = [] { const char* e = getenv("PARSEQ_TRAIN_SPLIT_TARGET"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 512; }();
    return t;
}

// asum (optional): [M] += row sums of A over k, folded into the product when it takes the bf16 matrix-core kernel; returns through
// *asum_done whether it did (the caller runs the column-sum kernel otherwise)
// bf16 shadow operands and outputs of a product (train_ops.h SgemmArgs a16 / b16 / c16 / gelu_out16): only on the matrix-core kernels of
// the bf16-operand mode; a product that asks for them and cannot take those kernels is an error, never a silent fp32 read of bf16 data
struct GemmExt {
    bool a16 = false, b16 = false;
    bf16_t* c16 = nullptr;
    bf16_t* gelu_out16 = nullptr;
    const bf16_t* gelu_pre16 = nullptr;
};
static int sgemm(const TrainCtx& cx, const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias, const float* R,
                 long ldr, int rper, float* C, long ldc, int M, int N, int K, float alpha, bool accumulate, float* asum = nullptr, bool* asum_done = nullptr,
                 const float* gelu_pre = nullptr, float* gelu_out = nullptr,      // gelu_pre / gelu_out: same contract as asum (folded on the bf16 matrix-core kernel, reported through asum_done)
                 const GemmExt* ext = nullptr) {
    hipStream_t s = cx.s;
    if (M <= 0 || N <= 0 || K <= 0) return fail(PARSEQ_E_INVALID, "sgemm: bad shape %d x %d x %d", M, N, K);
    SgemmArgs a{A, sam, sak, B, sbk, sbn, bias, R, ldr, rper > 0 ? rper : 1, C, ldc, M, N, K, alpha, accumulate ? 1 : 0, nullptr, nullptr, nullptr};
    if (asum_done) *asum_done = false;
    if (ext && (ext->a16 || ext->b16 || ext->c16 || ext->gelu_out16 || ext->gelu_pre16)) {
        if (!cx.bf16_ops || !cx.scratch || M < 16 || N < 16) return fail(PARSEQ_E_STATE, "sgemm: bf16 shadow operands outside the bf16-operand mode");
        if (!C && !ext->c16) return fail(PARSEQ_E_INVALID, "sgemm: no output");
        a.a16 = ext->a16; a.b16 = ext->b16; a.c16 = ext->c16; a.gelu_out16 = ext->gelu_out16; a.gelu_pre16 = ext->gelu_pre16;
        const int gm_ = (M + MG_BM - 1) / MG_BM, gn_ = (N + MG_BN - 1) / MG_BN, tiles = gm_ * gn_;
        const bool both = ext->a16 && ext->b16 && sak == 1 && sbk == 1;       // the 64-deep kernel
        const bool both_t = ext->a16 && ext->b16 && sam == 1 && sbn == 1;     // dW with a bf16 dY: both operands outer-contiguous
        const bool deep_t = both_t && K % BH_BK == 0;                       // ... at 64 rows of the contraction per stage
        const int bk = (both || deep_t) ? BH_BK : BG_BK;
        // alignment of the 16-byte (k-contiguous) / 8-byte (outer-contiguous) pieces the loaders read
        const bool a_ok16 = !ext->a16 ? (aligned16(A) && (sak == 1 ? sam % 4 == 0 : (sam == 1 && sak % 4 == 0 && M % 4 == 0)))
                                      : (aligned16(A) && (both ? sam % 8 == 0 : (both_t && sak % 4 == 0 && M % 4 == 0)));
        const bool b_ok16 = !ext->b16 ? (aligned16(B) && (sbk == 1 ? sbn % 4 == 0 : (sbn == 1 && sbk % 4 == 0 && N % 4 == 0)))
                                      : (aligned16(B) && (sbk == 1 ? sbn % 8 == 0 : (sbn == 1 && sbk % 4 == 0 && N % 4 == 0)));
        if (!a_ok16 || !b_ok16 || K % bk != 0 || (ext->a16 && !both && !both_t))
            return fail(PARSEQ_E_INVALID, "sgemm: shadow operands of a %d x %d x %d product are not laid out for the matrix-core kernels", M, N, K);
        int splits = 1;
        if (tiles < 256) {      // the same split as the fp32-in-memory path takes (32-deep stages), so that the two stay bit-identical
            splits = std::min((split_target() + tiles - 1) / tiles, K / (4 * BG_BK));
            splits = (int)std::min<size_t>((size_t)std::max(splits, 1), cx.scratch_floats / ((size_t)M * N + (size_t)M));
            splits = std::max(splits, 1);
        }
        const int k_chunk = ((K + splits - 1) / splits + bk - 1) / bk * bk;
        splits = (K + k_chunk - 1) / k_chunk;
        if (asum) { if (both) return fail(PARSEQ_E_INVALID, "sgemm: row sums of a bf16 shadow"); a.asum = asum; }
        a.gelu_pre = gelu_pre; a.gelu_out = gelu_out;
        if (asum_done) *asum_done = true;
        const dim3 grid_((unsigned)tiles, 1, splits);
        void (*kern)(const SgemmArgs, int, float*, int, int);
        // whole 128 x 128 tiles (every product of the PARSeq-S / ViTSTR encoders): the four-workgroups-per-CU forms (train_ops.h); the buffer
        // loads' 32-bit byte offsets cover both operands with room to spare at any batch that fits the workspace
        static const bool no_w4 = getenv("PARSEQ_TRAIN_GEMM_W3") != nullptr;
        const bool whole = M % MG_BM == 0 && N % MG_BN == 0 && !no_w4 &&
                           (size_t)M * (size_t)std::max(sam, sak) < ((size_t)1 << 29) && (size_t)N * (size_t)std::max(sbn, sbk) < ((size_t)1 << 29) &&
                           (size_t)K * (size_t)std::max(sak, sbk) < ((size_t)1 << 29);
        if (both) kern = whole ? mfma_bgemm16_kernel<true> : mfma_bgemm16_kernel<false>;
        else if (deep_t) kern = whole ? mfma_bgemm16t_kernel<true> : mfma_bgemm16t_kernel<false>;
        else if (both_t) kern = mfma_bgemm_kernel<false, false, true, true>;
        else if (ext->b16) kern = sak == 1 ? (sbk == 1 ? mfma_bgemm_kernel<true, true, true> : mfma_bgemm_kernel<true, false, true>)
                                           : (sbk == 1 ? mfma_bgemm_kernel<false, true, true> : mfma_bgemm_kernel<false, false, true>);
        else kern = sak == 1 ? (sbk == 1 ? mfma_bgemm_kernel<true, true, false> : mfma_bgemm_kernel<true, false, false>)
                             : (sbk == 1 ? mfma_bgemm_kernel<false, true, false> : mfma_bgemm_kernel<false, false, false>);
        hipLaunchKernelGGL(kern, grid_, dim3(256), 0, s, a, k_chunk, cx.scratch, gn_, gm_);
        HIPCHK(hipGetLastError());
        if (splits > 1) {
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * N + (a.asum ? (size_t)M : 0) + 255) / 256)), dim3(256), 0, s, a, cx.scratch, splits);
            HIPCHK(hipGetLastError());
        }
        return 0;
    }
    // matrix-core path: whole 128 x 128 tiles, whole 16-deep stages, 16-byte aligned rows along whichever axis is contiguous
    const bool a_ok = aligned16(A) && (sak == 1 ? sam % 4 == 0 : (sam == 1 && sak % 4 == 0));
    const bool b_ok = aligned16(B) && (sbk == 1 ? sbn % 4 == 0 : (sbn == 1 && sbk % 4 == 0));
    // bf16-operand mode: edge tiles allowed (the 95-class head, the 96-wide patch rows) as long as an outer-contiguous operand has whole
    // groups of four and at least one of them
    const bool bf16 = cx.bf16_ops && K % BG_BK == 0 && a_ok && b_ok && cx.scratch && M >= 16 && N >= 16 &&
                      (sak == 1 || M % 4 == 0) && (sbk == 1 || N % 4 == 0);
    if (bf16 || (M % MG_BM == 0 && N % MG_BN == 0 && K % MG_BK == 0 && a_ok && b_ok && cx.scratch)) {
        const int bk = bf16 ? BG_BK : MG_BK;
        const int gm_ = (M + MG_BM - 1) / MG_BM, gn_ = (N + MG_BN - 1) / MG_BN;
        const int tiles = gm_ * gn_;
        int splits = 1;
        if (tiles < 256) {
            splits = std::min((split_target() + tiles - 1) / tiles, K / (4 * bk));
            splits = (int)std::min<size_t>((size_t)std::max(splits, 1), cx.scratch_floats / ((size_t)M * N + (size_t)M));      // + the row-sum slots
            splits = std::max(splits, 1);
        }
        const int k_chunk = ((K + splits - 1) / splits + bk - 1) / bk * bk;
        splits = (K + k_chunk - 1) / k_chunk;
        if (bf16) {
            if (asum) { a.asum = asum; if (asum_done) *asum_done = true; }
            if (gelu_pre) { a.gelu_pre = gelu_pre; if (asum_done) *asum_done = true; }
            if (gelu_out) { a.gelu_out = gelu_out; if (asum_done) *asum_done = true; }
            const dim3 grid_((unsigned)(gn_ * gm_), 1, splits);      // one-dimensional tile index: the kernel orders the tiles XCD-aware
            if (sak == 1 && sbk == 1) hipLaunchKernelGGL((mfma_bgemm_kernel<true, true>), grid_, dim3(256), 0, s, a, k_chunk, cx.scratch, gn_, gm_);
            else if (sak == 1) hipLaunchKernelGGL((mfma_bgemm_kernel<true, false>), grid_, dim3(256), 0, s, a, k_chunk, cx.scratch, gn_, gm_);
            else if (sbk == 1) hipLaunchKernelGGL((mfma_bgemm_kernel<false, true>), grid_, dim3(256), 0, s, a, k_chunk, cx.scratch, gn_, gm_);
            else hipLaunchKernelGGL((mfma_bgemm_kernel<false, false>), grid_, dim3(256), 0, s, a, k_chunk, cx.scratch, gn_, gm_);
        } else
        hipLaunchKernelGGL(mfma_sgemm_kernel, dim3(N / MG_BN, M / MG_BM, splits), dim3(256), 0, s, a, k_chunk, cx.scratch);
        HIPCHK(hipGetLastError());
        if (splits > 1) {
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * N + (a.asum ? (size_t)M : 0) + 255) / 256)), dim3(256), 0, s, a, cx.scratch, splits);
            HIPCHK(hipGetLastError());
        }
        return 0;
    }
    hipLaunchKernelGGL(sgemm_kernel, dim3((N + SG_BN - 1) / SG_BN, (M + SG_BM - 1) / SG_BM), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}
static int colsum(const TrainCtx& cx, const float* A, long lda, int M, int N, float* out, bool accumulate) {
    hipStream_t s = cx.s;
    constexpr int CHUNKS = 64;
    if (M >= 2048 && cx.scratch && (size_t)CHUNKS * N <= cx.scratch_floats) {
        const int rows_per = (M + CHUNKS - 1) / CHUNKS;
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, CHUNKS), dim3(1024), 0, s, A, lda, M, N, cx.scratch, 0, rows_per, (float*)nullptr, 0);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, 1), dim3(1024), 0, s, cx.scratch, (long)N, CHUNKS, N, out, accumulate ? 1 : 0, CHUNKS, (float*)nullptr, 0);
        HIPCHK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, 1), dim3(1024), 0, s, A, lda, M, N, out, accumulate ? 1 : 0, M, (float*)nullptr, 0);
    HIPCHK(hipGetLastError());
    return 0;
}
// y[M, N] = x[M, K] W[N, K]^T + bias + R[m % rper]
// gelu_out (optional, [M, N]): gelu(y) as a second output — from the product's epilogue on the bf16 matrix-core kernel, by gelu_fwd_kernel otherwise
static int lin_fwd(const TrainCtx& cx, const float* x, const float* W, const float* bias, const float* R, int rper, float* y, int M, int N, int K,
                   float* gelu_out = nullptr) {
    bool fused = false;
    CHK(sgemm(cx, x, K, 1, W, 1, K, bias, R, N, rper, y, N, M, N, K, 1.f, false, nullptr, &fused, nullptr, gelu_out));
    if (gelu_out && !fused) {
        hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)(((size_t)M * N + 1023) / 1024)), dim3(256), 0, cx.s, y, gelu_out, (size_t)M * N);
        HIPCHK(hipGetLastError());
    }
    return 0;
}
// dW[N, K] += dy[M, N]^T x[M, K];  db[N] += column sums of dy;  dx[M, K] = dy W   (dx may be null)
// dx_gelu_pre (optional, [M, K]): dx is additionally multiplied by gelu'(dx_gelu_pre) — the GELU backward of the layer below, folded
// into the dX product's epilogue when it takes the bf16 matrix-core kernel and run as gelu_bwd_kernel otherwise
static int lin_bwd(const TrainCtx& cx, const float* x, const float* W, const float* dy, float* dW, float* db, float* dx, int M, int N, int K,
                   const float* dx_gelu_pre = nullptr) {
    // bf16-operand mode, output width not a multiple of 4 (the 95-class head): rows of dy are not 16-byte aligned and N is no multiple of
    // the 32-deep k-step, so both products would fall to the VALU kernel (10 % of the step).  Instead dy and W are copied into zero-padded
    // [M, Np] / [Np, K] buffers (Np = N rounded up to 32) carved off the end of the scratch, both products run on the matrix cores, and the
    // first N rows of the padded dW are added to the gradient.
    const int Np = (N + 31) / 32 * 32;
    const size_t reserve = (size_t)M * Np + 2 * (size_t)Np * K;
    if (cx.bf16_ops && N % 4 != 0 && cx.scratch && M % 4 == 0 && K % 4 == 0 && reserve + ((size_t)4 << 20) <= cx.scratch_floats) {
        hipStream_t s = cx.s;
        TrainCtx c2 = cx; c2.scratch_floats = cx.scratch_floats - reserve;
        float* dyp = cx.scratch + c2.scratch_floats; float* Wp = dyp + (size_t)M * Np; float* dWp = Wp + (size_t)Np * K;
        hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)(((size_t)M * Np + 255) / 256)), dim3(256), 0, s, dy, M, N, dyp, M, Np);
        hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)(((size_t)Np * K + 255) / 256)), dim3(256), 0, s, W, N, K, Wp, Np, K);
        HIPCHK(hipGetLastError());
        CHK(sgemm(c2, dyp, 1, Np, x, K, 1, nullptr, nullptr, 0, 0, dWp, K, Np, K, M, 1.f, false));
        hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, s, dWp, dW, (size_t)N * K);
        HIPCHK(hipGetLastError());
        CHK(colsum(c2, dy, N, M, N, db, true));
        if (dx) CHK(sgemm(c2, dyp, Np, 1, Wp, K, 1, nullptr, nullptr, 0, 0, dx, K, M, K, Np, 1.f, false));
        if (dx && dx_gelu_pre) { hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)(((size_t)M * K + 1023) / 1024)), dim3(256), 0, s, dx_gelu_pre, dx, dx, (size_t)M * K); HIPCHK(hipGetLastError()); }
        return 0;
    }
    // dW += dY^T X; the bias gradient (column sums of dY = row sums of the product's A operand) rides on it in the bf16-operand mode
    bool db_done = false;
    CHK(sgemm(cx, dy, 1, N, x, K, 1, nullptr, nullptr, 0, 0, dW, K, N, K, M, 1.f, true, db, &db_done));
    if (!db_done) CHK(colsum(cx, dy, N, M, N, db, true));
    if (dx) {
        bool fused = false;
        CHK(sgemm(cx, dy, N, 1, W, K, 1, nullptr, nullptr, 0, 0, dx, K, M, K, N, 1.f, false, nullptr, &fused, dx_gelu_pre));
        if (dx_gelu_pre && !fused) {
            hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)(((size_t)M * K + 1023) / 1024)), dim3(256), 0, cx.s, dx_gelu_pre, dx, dx, (size_t)M * K);
            HIPCHK(hipGetLastError());
        }
    }
    return 0;
}
// The same three products on bf16 SHADOW operands (encoder, bf16-operand mode; train_ops.h SgemmArgs): x16 [M, K] and the weight shadows W16 [N, K] /
// Wt16 [K, N] are bfloat16 in memory; dy stays fp32 where it is the A operand of the dW product (the bias gradient is summed from the
// unrounded values there) and is read through its shadow dy16 (when the producer wrote one) by the dX product; dx16 / gelu_out16: the
// result again as bf16 for the next product.  Bit-identical to lin_fwd / lin_bwd on the fp32 copies: the rounding moved, nothing else.
// y may be nullptr when y16 is given (bf16-only storage of the result)
static int lin_fwd16(const TrainCtx& cx, const bf16_t* x16, const bf16_t* W16, const float* bias, const float* R, int rper, float* y, int M, int N, int K,
                     bf16_t* gelu_out16 = nullptr, bf16_t* y16 = nullptr) {
    GemmExt e; e.a16 = e.b16 = true; e.gelu_out16 = gelu_out16; e.c16 = y16;
    return sgemm(cx, reinterpret_cast<const float*>(x16), K, 1, reinterpret_cast<const float*>(W16), 1, K, bias, R, N, rper, y, N, M, N, K, 1.f, false,
                 nullptr, nullptr, nullptr, nullptr, &e);
}
// dy may be nullptr when dy16 is given (the gradient exists as bf16 only: both products read it, the bias gradient sums the bf16 values);
// dx may be nullptr when dx16 is given; dx_gelu_pre16: the pre-activation as bf16
static int lin_bwd16_dw(const TrainCtx& cx, const bf16_t* x16, const float* dy, const bf16_t* dy16, float* dW, float* db, int M, int N, int K) {
    if (!dy && !dy16) return fail(PARSEQ_E_INVALID, "lin_bwd16: no gradient");
    GemmExt ew; ew.b16 = true; ew.a16 = dy == nullptr;
    return sgemm(cx, dy ? dy : reinterpret_cast<const float*>(dy16), 1, N, reinterpret_cast<const float*>(x16), K, 1, nullptr, nullptr, 0, 0, dW, K, N, K, M, 1.f, true,
                 db, nullptr, nullptr, nullptr, &ew);
}
static int lin_bwd16_dx(const TrainCtx& cx, const bf16_t* Wt16, const float* dy, const bf16_t* dy16, float* dx, bf16_t* dx16, int M, int N, int K,
                        const float* dx_gelu_pre = nullptr, const bf16_t* dx_gelu_pre16 = nullptr) {
    if (!dy && !dy16) return fail(PARSEQ_E_INVALID, "lin_bwd16: no gradient");
    if (!dx && !dx16) return 0;
    GemmExt ex; ex.b16 = true; ex.a16 = dy16 != nullptr; ex.c16 = dx16; ex.gelu_pre16 = dx_gelu_pre16;
    return sgemm(cx, dy16 ? reinterpret_cast<const float*>(dy16) : dy, N, 1, reinterpret_cast<const float*>(Wt16), 1, N, nullptr, nullptr, 0, 0, dx, K, M, K, N,
                 1.f, false, nullptr, nullptr, dx_gelu_pre, nullptr, &ex);
}
static int lin_bwd16(const TrainCtx& cx, const bf16_t* x16, const bf16_t* Wt16, const float* dy, const bf16_t* dy16, float* dW, float* db, float* dx,
                     bf16_t* dx16, int M, int N, int K, const float* dx_gelu_pre = nullptr, const bf16_t* dx_gelu_pre16 = nullptr) {
    CHK(lin_bwd16_dw(cx, x16, dy, dy16, dW, db, M, N, K));
    return lin_bwd16_dx(cx, Wt16, dy, dy16, dx, dx16, M, N, K, dx_gelu_pre, dx_gelu_pre16);
}
// dx = add + LayerNorm backward; dgamma += column sums of dy * xhat; dbeta += column sums of dy.  `tmp` is [rows, E] scratch.
// dx16 (optional): dx again as bf16, the operand shadow of the dX product that follows.
static int ln_bwd(const TrainCtx& cx, const float* x, const float* gamma, const float* dy, const float* add, float* dx, float* dgamma, float* dbeta,
                  float* tmp, int rows, int E, float eps, bf16_t* dx16 = nullptr) {
    hipStream_t s = cx.s;
    if (E > 768) return fail(PARSEQ_E_INVALID, "layernorm backward: E=%d > 768", E);
    // per-chunk partial sums of dy * xhat and dy land in the scratch ([chunks][2E]); two small column sums fold them (`tmp` is no longer used)
    (void)tmp;
    const int chunks = (rows + LNB_ROWS - 1) / LNB_ROWS;
    if (!cx.scratch || (size_t)chunks * 2 * E + (size_t)64 * E > cx.scratch_floats) return fail(PARSEQ_E_INVALID, "layernorm backward: %d rows do not fit the scratch", rows);
    float* part = cx.scratch + (cx.scratch_floats - (size_t)chunks * 2 * E);      // the END of the scratch: colsum's own partials use its start
    TrainCtx c2 = cx; c2.scratch_floats = cx.scratch_floats - (size_t)chunks * 2 * E;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(chunks), dim3(256), 0, s, x, gamma, dy, add, dx, part, rows, E, eps, dx16);
    HIPCHK(hipGetLastError());
    // both column sums in one pair of launches: [chunks][2E] -> 64 row chunks -> dgamma (columns < E) and dbeta (the rest)
    constexpr int CHUNKS = 64;
    if (chunks >= 2048 && (size_t)CHUNKS * 2 * E <= c2.scratch_floats) {
        const int rows_per = (chunks + CHUNKS - 1) / CHUNKS;
        hipLaunchKernelGGL(colsum_kernel, dim3((2 * E + 63) / 64, CHUNKS), dim3(1024), 0, s, part, 2L * E, chunks, 2 * E, c2.scratch, 0, rows_per, (float*)nullptr, 0);
        hipLaunchKernelGGL(colsum_kernel, dim3((2 * E + 63) / 64, 1), dim3(1024), 0, s, c2.scratch, 2L * E, CHUNKS, 2 * E, dgamma, 1, CHUNKS, dbeta, E);
        HIPCHK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((2 * E + 63) / 64, 1), dim3(1024), 0, s, part, 2L * E, chunks, 2 * E, dgamma, 1, chunks, dbeta, E);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int HD>
static int train_attn_hd(const TrainCtx& cx, const TrainAttnArgs& a, int B, bool backward) {
    hipStream_t s = cx.s;
    const size_t lds = train_attn_lds_floats(a.Lq, a.Lk, HD, backward) * sizeof(float);
    if (lds > 150 * 1024 || (size_t)a.Lk * HD > (size_t)TA_NACC * 256)
        return fail(PARSEQ_E_INVALID, "training attention: %d keys of width %d do not fit (LDS %zu bytes)", a.Lk, HD, lds);
    static LdsAttr attr_f, attr_b;      // one pair per head width
    HIPCHK(attr_f.ensure(reinterpret_cast<const void*>(train_attn_kernel<false, HD>), 150 * 1024));
    HIPCHK(attr_b.ensure(reinterpret_cast<const void*>(train_attn_kernel<true, HD>), 150 * 1024));
    if (backward) hipLaunchKernelGGL((train_attn_kernel<true, HD>), dim3(B * a.H), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((train_attn_kernel<false, HD>), dim3(B * a.H), dim3(256), lds, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}
// encoder shape on the matrix cores (train_attn_mfma_kernel): head width 64, whole 32-row query blocks and 16-key tiles, no masks
static int train_attn_mfma(const TrainCtx& cx, const TrainAttnArgs& a, int B, bool backward) {
    hipStream_t s = cx.s;
    const size_t lds = ((size_t)2 * a.Lk * 65 + (size_t)(backward ? 2 : 1) * 32 * 65 + (size_t)(backward ? 2 : 1) * 32 * (a.Lk + 1)) * sizeof(float);
    static LdsAttr attr_f, attr_b;
    HIPCHK(attr_f.ensure(reinterpret_cast<const void*>(train_attn_mfma_kernel<false>), 150 * 1024));
    HIPCHK(attr_b.ensure(reinterpret_cast<const void*>(train_attn_mfma_kernel<true>), 150 * 1024));
    if (backward) hipLaunchKernelGGL((train_attn_mfma_kernel<true>), dim3(B * a.H), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((train_attn_mfma_kernel<false>), dim3(B * a.H), dim3(256), lds, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}
// encoder shape in the bf16-operand mode (train_attn_bf16_kernel): 128 tokens, head width 64, per-image queries, no masks, no dropout
static int train_attn_bf16(const TrainCtx& cx, const TrainAttnArgs& a, int B, bool backward) {
    hipStream_t s = cx.s;
    static LdsAttr attr_f, attr_b;
    HIPCHK(attr_f.ensure(reinterpret_cast<const void*>(train_attn_bf16_kernel<false>), train_attn_bf16_lds(false)));
    HIPCHK(attr_b.ensure(reinterpret_cast<const void*>(train_attn_bf16_kernel<true>), train_attn_bf16_lds(true)));
    if (backward) hipLaunchKernelGGL((train_attn_bf16_kernel<true>), dim3(B * a.H), dim3(256), train_attn_bf16_lds(true), s, a);
    else hipLaunchKernelGGL((train_attn_bf16_kernel<false>), dim3(B * a.H), dim3(256), train_attn_bf16_lds(false), s, a);
    HIPCHK(hipGetLastError());
    return 0;
}
// decoder shapes in the bf16-operand mode (train_attn_dec_bf16_kernel): head width 32, <= 32 queries, <= 128 keys, masks, dropout
static int train_attn_dec_bf16(const TrainCtx& cx, const TrainAttnArgs& a, int B, bool backward) {
    hipStream_t s = cx.s;
    static LdsAttr attr_f, attr_b, attr_f2, attr_b2;
    HIPCHK(attr_f.ensure(reinterpret_cast<const void*>(train_attn_dec_bf16_kernel<false, 8>), train_attn_dec_lds(false, 8)));
    HIPCHK(attr_b.ensure(reinterpret_cast<const void*>(train_attn_dec_bf16_kernel<true, 8>), train_attn_dec_lds(true, 8)));
    HIPCHK(attr_f2.ensure(reinterpret_cast<const void*>(train_attn_dec_bf16_kernel<false, 2>), train_attn_dec_lds(false, 2)));
    HIPCHK(attr_b2.ensure(reinterpret_cast<const void*>(train_attn_dec_bf16_kernel<true, 2>), train_attn_dec_lds(true, 2)));
    if (a.pass_loop > 1 && !(a.pass_B > 0 && a.kv_shared && B == a.pass_B * a.pass_loop))
        return fail(PARSEQ_E_INVALID, "training attention: pass_loop needs pass_B, shared K / V and a batch of pass_B * pass_loop images");
    const int blocks = (a.pass_loop > 1 ? a.pass_B : B) * a.H;      // pass_loop: one workgroup per (image, head) walks the passes
    const bool small = a.Lk <= 32 && !getenv("PARSEQ_TRAIN_ATTN_KT8");      // the self-attention: the 32-key instantiation (a quarter of the LDS, a third of the registers)
    if (small) {
        if (backward) hipLaunchKernelGGL((train_attn_dec_bf16_kernel<true, 2>), dim3(blocks), dim3(128), train_attn_dec_lds(true, 2), s, a);
        else hipLaunchKernelGGL((train_attn_dec_bf16_kernel<false, 2>), dim3(blocks), dim3(128), train_attn_dec_lds(false, 2), s, a);
    } else {
        if (backward) hipLaunchKernelGGL((train_attn_dec_bf16_kernel<true, 8>), dim3(blocks), dim3(128), train_attn_dec_lds(true, 8), s, a);
        else hipLaunchKernelGGL((train_attn_dec_bf16_kernel<false, 8>), dim3(blocks), dim3(128), train_attn_dec_lds(false, 8), s, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
// whether a decoder-shaped attention call (forward and backward) runs on train_attn_dec_bf16_kernel
static bool train_attn_is_dec_bf16(const TrainCtx& cx, const TrainAttnArgs& a, int hd) {
    return cx.bf16_ops && hd == TD_HD && a.Lq <= TD_Q && a.Lk <= TD_K && a.ldq % 4 == 0 && a.ldkv % 4 == 0 && a.ldo % 4 == 0 && a.q_bstride % 4 == 0 &&
           a.lddq % 4 == 0 && a.lddkv % 4 == 0 && aligned16(a.dk) && aligned16(a.dv) && !getenv("PARSEQ_TRAIN_F32_ATTN");
}
static int train_attn(const TrainCtx& cx, const TrainAttnArgs& a, int B, bool backward, int hd) {
    if (cx.bf16_ops && hd == TD_HD && a.Lq <= TD_Q && a.Lk <= TD_K && a.ldq % 4 == 0 && a.ldkv % 4 == 0 && a.ldo % 4 == 0 && a.q_bstride % 4 == 0 &&
        (!backward || (a.lddq % 4 == 0 && a.lddkv % 4 == 0 && aligned16(a.dk) && aligned16(a.dv))) && !getenv("PARSEQ_TRAIN_F32_ATTN"))
        return train_attn_dec_bf16(cx, a, B, backward);
    if (a.pass_loop > 1) return fail(PARSEQ_E_INVALID, "training attention: pass_loop is train_attn_dec_bf16_kernel's alone");
    if (a.pass_B && hd != TD_HD) return fail(PARSEQ_E_INVALID, "training attention: several passes per launch only at the decoder's head width");
    if (cx.bf16_ops && hd == TB_HD && a.Lq == TB_N && a.Lk == TB_N && !a.qmask && !a.kmask && !a.drop.thresh && a.q_bstride == (long)a.Lq * a.ldq &&
        a.ldq % 4 == 0 && a.ldkv % 4 == 0 && a.ldo % 4 == 0 && (!backward || (a.lddq % 4 == 0 && a.lddkv % 4 == 0 && aligned16(a.dk) && aligned16(a.dv))) &&
        !getenv("PARSEQ_TRAIN_F32_ATTN"))
        return train_attn_bf16(cx, a, B, backward);
    if (a.o16 || a.dq16) return fail(PARSEQ_E_INVALID, "training attention: a bf16 output is only written by the encoder-shaped bf16 kernel");
    if (hd == 64 && a.Lq % 32 == 0 && a.Lk % 16 == 0 && a.Lk <= 128 && !a.qmask && !a.kmask && !a.drop.thresh && !getenv("PARSEQ_TRAIN_VALU_ATTN"))
        return train_attn_mfma(cx, a, B, backward);
    if (hd == 32) return train_attn_hd<32>(cx, a, B, backward);
    if (hd == 64) return train_attn_hd<64>(cx, a, B, backward);
    return fail(PARSEQ_E_INVALID, "training attention: head width %d not in {32, 64}", hd);
}

// The K permutation passes of a step share every weight and differ in their masks, dropout sites and (after two passes) targets only
// (system.py:175-196), so the decoder runs them as ONE batch of KP * B images (KP = K by default): every Linear product, LayerNorm,
// attention launch and column sum once per step instead of once per pass — 6 x the rows per launch, a sixth of the launches and of the
// split-K folds, the dW products contracted over all passes at once.  PARSEQ_TRAIN_PERM_GROUP=g runs the passes g at a time (1 = one after
// the other, the arrangement of rounds 1-2: same masks, same per-pass losses, gradients equal up to fp32 summation order).
static int train_perm_group(int K) {
    int g = K;
    if (const char* e = getenv("PARSEQ_TRAIN_PERM_GROUP")) { const int v = atoi(e); if (v >= 1) g = v; }
    return std::min(std::max(g, 1), K);
}
struct TrainDecoderLayout {          // offsets in floats into the caller's workspace
    size_t content0, content, cn, kvc, qd, qn, qsa, kvm, sa_o, t1, n1, q2, ca_o, t2, n2, hpre, hact, t3, out, logits;
    size_t d_a, d_b, d_c, d_h, pm, d_kvc, d_kvm, d_kvm_p, d_content, d_pq, d_qb, row_loss, tgt_all, losses, counts, scratch, scratch_floats, total;
    int KP;                          // passes per batch
    bool ca_loop;                    // the cross-attention walks the passes of a batch inside one workgroup (no per-pass d K | d V copies)
};
static TrainDecoderLayout train_decoder_layout(const parseq_model* m, int B, int L, int K) {
    const size_t E = m->cfg.embed_dim, F = E * m->cfg.dec_mlp_ratio, S = m->tokens, C = m->classes, M = (size_t)B * L, MS = (size_t)B * S;
    TrainDecoderLayout o;
    o.KP = train_perm_group(K);
    const size_t P = (size_t)o.KP, MP = P * M;      // rows of a per-pass buffer
    size_t off = 0;
    auto take = [&](size_t n) { const size_t at = off; off += (n + 63) / 64 * 64; return at; };
    o.content0 = take(M * E); o.content = take(MP * E); o.cn = take(MP * E); o.kvc = take(MP * 2 * E); o.qd = take(MP * E); o.qn = take(MP * E);
    o.qsa = take(MP * E); o.kvm = take(MS * 2 * E);
    o.sa_o = take(MP * E); o.t1 = take(MP * E); o.n1 = take(MP * E); o.q2 = take(MP * E); o.ca_o = take(MP * E); o.t2 = take(MP * E); o.n2 = take(MP * E);
    o.hpre = take(MP * F); o.hact = take(MP * F); o.t3 = take(MP * E); o.out = take(MP * E); o.logits = take(MP * C);
    o.d_a = take(MP * E); o.d_b = take(MP * E); o.d_c = take(MP * E); o.d_h = take(MP * F); o.pm = take(MP * E);
    o.d_kvc = take(MP * 2 * E); o.d_kvm = take(MS * 2 * E);
    // each pass's own d K | d V of the memory, folded into d_kvm after the batch — only where the cross-attention cannot walk the passes itself
    // (train_attn_dec_bf16_kernel's pass_loop: bf16-operand mode, head width 32, <= TD_Q queries, <= TD_K memory tokens)
    o.ca_loop = o.KP > 1 && m->train_precision == PARSEQ_BF16 && E == (size_t)m->cfg.dec_heads * TD_HD && L <= TD_Q && (int)S <= TD_K && E % 4 == 0 &&
                !getenv("PARSEQ_TRAIN_F32_ATTN") && !getenv("PARSEQ_TRAIN_NO_PASS_LOOP");
    o.d_kvm_p = (o.KP > 1 && !o.ca_loop) ? take(P * MS * 2 * E) : o.d_kvm;
    o.d_content = take(M * E); o.d_pq = take(L * E); o.d_qb = take(MP * E);
    o.row_loss = take(MP); o.tgt_all = take((size_t)K * M); o.losses = take(K + 1); o.counts = take(K + 1);
    o.scratch_floats = train_scratch_floats(MP, E); o.scratch = take(o.scratch_floats);
    o.total = off;
    return o;
}

extern "C" int64_t parseq_model_param_offset(const parseq_model* m, int index) {
    if (!m || index < 0 || index >= (int)m->params.size()) return -1;
    return (int64_t)m->params[index].offset;
}
extern "C" int64_t parseq_model_grad_elems(const parseq_model* m) { return m ? (int64_t)m->master_elems : 0; }
extern "C" int parseq_model_set_train_precision(parseq_model* m, int precision) {
    if (!m) return fail(PARSEQ_E_INVALID, "null model");
    if (precision != PARSEQ_F32 && precision != PARSEQ_BF16) return fail(PARSEQ_E_INVALID, "training precision %d (PARSEQ_F32 or PARSEQ_BF16)", precision);
    m->train_precision = precision;
    return 0;
}

// Where a named intermediate of the LAST permutation (or an accumulator) lives in the workspace, in floats; -1 if unknown.  For tests.
extern "C" int64_t parseq_train_decoder_workspace_offset(const parseq_model* m, int batch, int ctx_len, int num_perms, const char* name) {
    if (!m || !name || batch <= 0 || ctx_len <= 0 || num_perms <= 0) return -1;
    const TrainDecoderLayout o = train_decoder_layout(m, batch, ctx_len, num_perms);
    const size_t E = m->cfg.embed_dim, F = E * m->cfg.dec_mlp_ratio, C = m->classes, M = (size_t)batch * ctx_len;
    const size_t last = (size_t)((num_perms - 1) % o.KP);      // the last pass's slot in its batch of KP passes
    struct Entry { const char* name; size_t off, width; };      // width: floats per row of a per-pass buffer; 0 = shared by the passes
    const Entry table[] = {
        {"content", o.content, E}, {"cn", o.cn, E}, {"kvc", o.kvc, 2 * E}, {"qd", o.qd, E}, {"qn", o.qn, E}, {"qsa", o.qsa, E}, {"kvm", o.kvm, 0},
        {"sa_o", o.sa_o, E}, {"t1", o.t1, E}, {"n1", o.n1, E}, {"q2", o.q2, E}, {"ca_o", o.ca_o, E}, {"t2", o.t2, E}, {"n2", o.n2, E},
        {"hpre", o.hpre, F}, {"hact", o.hact, F}, {"t3", o.t3, E}, {"out", o.out, E}, {"dlogits", o.logits, C}, {"d_kvc", o.d_kvc, 2 * E},
        {"d_kvm", o.d_kvm, 0}, {"d_content", o.d_content, 0}, {"d_pq", o.d_pq, 0}};
    for (const Entry& e : table) if (!strcmp(e.name, name)) return (int64_t)(e.off + last * M * e.width);
    return -1;
}

extern "C" size_t parseq_train_decoder_workspace_bytes(const parseq_model* m, int batch, int ctx_len, int num_perms) {
    if (!m || batch <= 0 || ctx_len <= 0 || num_perms <= 0) return 0;
    return train_decoder_layout(m, batch, ctx_len, num_perms).total * sizeof(float);
}

// y = R + dropout(x) over `passes` passes of n_pass elements each (train_ops.h dropout_passes_kernel: R may be null, x == y allowed,
// x_shared: one pass of x read by every pass); with dropout off a plain add / copy
static int dropout_add(const TrainCtx& cx, const float* x, bool x_shared, const float* R, float* y, size_t n_pass, int passes, const DropSpec& d, unsigned site) {
    hipStream_t s = cx.s;
    hipLaunchKernelGGL(dropout_passes_kernel, dim3((unsigned)((n_pass + 255) / 256), (unsigned)passes), dim3(256), 0, s, x, x_shared ? 1 : 0, R, y, n_pass, d, site);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int parseq_train_decoder(parseq_model* m, const float* memory, const int32_t* tokens, const int32_t* targets, const uint8_t* key_padding_mask,
                                    const uint8_t* query_masks, int batch, int ctx_len, int num_perms, int total_targets, float dropout_p,
                                    uint64_t seed, float* loss_out, float* grads, float* dmemory, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !memory || !tokens || !targets || !key_padding_mask || !query_masks || !loss_out || !grads || !dmemory || !workspace)
        return fail(PARSEQ_E_INVALID, "null argument");
    if (m->vitstr) return fail(PARSEQ_E_INVALID, "ViTSTR has no decoder");
    for (const ParamSpec& ps : m->params) if (!ps.set) return fail(PARSEQ_E_STATE, "parameter %s was never set", ps.key.c_str());
    m->grad_events_valid = false;      // a new step starts writing the flat gradient buffer: the previous step's segment events say nothing about it
    DevGuard dg(m->device);
    const int B = batch, L = ctx_len, K = num_perms;
    if (B <= 0 || L < 2 || L > m->cfg.max_label_length + 1 || K <= 0 || total_targets <= 0)
        return fail(PARSEQ_E_INVALID, "bad shape: batch %d, ctx_len %d (2..%d), %d permutations, %d targets", B, L, m->cfg.max_label_length + 1, K, total_targets);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(PARSEQ_E_INVALID, "dropout_p %g outside [0, 1)", dropout_p);
    const TrainDecoderLayout o = train_decoder_layout(m, B, L, K);
    if (workspace_bytes < o.total * sizeof(float)) return fail(PARSEQ_E_INVALID, "workspace: %zu bytes given, %zu needed", workspace_bytes, o.total * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    const int E = m->cfg.embed_dim, F = E * m->cfg.dec_mlp_ratio, S = m->tokens, C = m->classes, H = m->cfg.dec_heads, M = B * L, MS = B * S;
    const float eps = m->cfg.dec_ln_eps, scale = 1.0f / sqrtf(32.0f), sqrtE = sqrtf((float)E);
    DropSpec drop{(unsigned)(seed & 0xFFFFFFFFull), (unsigned)(seed >> 32), 0u, 1.0f};
    if (dropout_p > 0.f) { drop.thresh = (unsigned)((double)dropout_p * 4294967296.0); drop.scale = 1.0f / (1.0f - dropout_p); }
    float* w = reinterpret_cast<float*>(workspace);
    const std::string p = "decoder.layers.0.";
    auto P = [&](const std::string& key) { return m->p(key); };
    auto G = [&](const std::string& key) { return grads + m->params[m->index.at(key)].offset; };
    const float* pq = P("pos_queries");
    const float* sa_w = P(p + "self_attn.in_proj_weight"); const float* sa_b = P(p + "self_attn.in_proj_bias");
    const float* ca_w = P(p + "cross_attn.in_proj_weight"); const float* ca_b = P(p + "cross_attn.in_proj_bias");
    float* content0 = w + o.content0; float* content = w + o.content; float* cn = w + o.cn; float* kvc = w + o.kvc; float* qd = w + o.qd;
    float* qn = w + o.qn; float* qsa = w + o.qsa; float* kvm = w + o.kvm;
    float* sa_o = w + o.sa_o; float* t1 = w + o.t1; float* n1 = w + o.n1; float* q2 = w + o.q2; float* ca_o = w + o.ca_o; float* t2 = w + o.t2;
    float* n2 = w + o.n2; float* hpre = w + o.hpre; float* hact = w + o.hact; float* t3 = w + o.t3; float* out = w + o.out; float* logits = w + o.logits;
    float* d_a = w + o.d_a; float* d_b = w + o.d_b; float* d_c = w + o.d_c; float* d_h = w + o.d_h; float* pm = w + o.pm;
    float* d_kvc = w + o.d_kvc; float* d_kvm = w + o.d_kvm; float* d_kvm_p = w + o.d_kvm_p; float* d_content = w + o.d_content; float* d_pq = w + o.d_pq;
    float* d_qb = w + o.d_qb;
    float* row_loss = w + o.row_loss; int* tgt_all = reinterpret_cast<int*>(w + o.tgt_all); float* losses = w + o.losses; int* counts = reinterpret_cast<int*>(w + o.counts);
    const size_t ME = (size_t)M * E, MF = (size_t)M * F;
    const int KP = o.KP;                           // passes per batch (train_perm_group)
    const TrainCtx cx{s, w + o.scratch, m->train_precision == PARSEQ_BF16, o.scratch_floats};

    // ---- shared by all permutations: the content rows before dropout, and the memory's K / V (model.py:95-98, modules.py:74) ----
    hipLaunchKernelGGL(train_content_kernel, dim3(M), dim3(256), 0, s, P("text_embed.embedding.weight"), pq, tokens, L, L, E, sqrtE, content0);
    HIPCHK(hipGetLastError());
    CHK(lin_fwd(cx, memory, ca_w + (size_t)E * E, ca_b + E, nullptr, 0, kvm, MS, 2 * E, E));
    if (KP == 1) HIPCHK(hipMemsetAsync(d_kvm, 0, (size_t)MS * 2 * E * sizeof(float), s));      // the passes accumulate into it one after the other
    HIPCHK(hipMemsetAsync(d_pq, 0, (size_t)L * E * sizeof(float), s));
    // the targets of pass i, one row per pass: <eos> targets are dropped after two permutations (system.py:191-195)
    for (int i = 0; i < K; ++i)
        HIPCHK(hipMemcpyAsync(tgt_all + (size_t)i * M, targets + (size_t)(i < 2 ? 0 : 1) * M, (size_t)M * sizeof(int), hipMemcpyDeviceToDevice, s));

    TrainAttnArgs sa{};      // self-attention of the query stream over the content stream (modules.py:70-72)
    sa.q = qsa; sa.q_bstride = (long)L * E; sa.ldq = E; sa.k = kvc; sa.v = kvc + E; sa.ldkv = 2 * E; sa.kmask = key_padding_mask; sa.ldkm = L;
    sa.o = sa_o; sa.ldo = E; sa.d_o = d_b; sa.dq = d_qb; sa.lddq = E; sa.dk = d_kvc; sa.dv = d_kvc + E; sa.lddkv = 2 * E;
    sa.Lq = L; sa.Lk = L; sa.H = H; sa.scale = scale; sa.kv_accumulate = 0; sa.drop = drop;
    sa.pass_B = B; sa.qmask_pstride = (long)L * L; sa.site_pstride = 8; sa.kv_shared = 0;
    TrainAttnArgs ca{};      // cross-attention over the encoder memory (modules.py:74-75)
    ca.q = q2; ca.q_bstride = (long)L * E; ca.ldq = E; ca.k = kvm; ca.v = kvm + E; ca.ldkv = 2 * E; ca.o = ca_o; ca.ldo = E; ca.d_o = d_c;
    ca.dq = d_a; ca.lddq = E; ca.dk = d_kvm_p; ca.dv = d_kvm_p + E; ca.lddkv = 2 * E; ca.Lq = L; ca.Lk = S; ca.H = H; ca.scale = scale;
    ca.kv_accumulate = KP == 1 ? 1 : 0;      // KP == 1: d_kvm_p IS d_kvm; otherwise each pass of the batch writes its own copy
    ca.drop = drop;
    ca.pass_B = B; ca.qmask_pstride = 0; ca.site_pstride = 8; ca.kv_shared = 1;
    // bf16-operand mode: one workgroup per (image, head) walks the batch's passes (train_ops.h TrainAttnArgs::pass_loop) — the memory's K | V
    // are staged once per batch instead of once per pass and d K | d V go straight into d_kvm, summed over the passes in the accumulators
    const bool ca_loop = o.ca_loop;      // decided by the layout (which then has no per-pass copies); the kernel's own preconditions must agree
    if (ca_loop) { ca.dk = d_kvm; ca.dv = d_kvm + E; }
    if (ca_loop && !train_attn_is_dec_bf16(cx, ca, 32)) return fail(PARSEQ_E_STATE, "training decoder: the workspace was laid out for the pass-walking cross-attention, which this call cannot run (alignment or environment changed)");
    enum { S_CONTENT, S_QUERY, S_SA_PROB, S_SA_OUT, S_CA_PROB, S_CA_OUT, S_FF_HIDDEN, S_FF_OUT };      // dropout sites of one pass

    for (int i0 = 0; i0 < K; i0 += KP) {
        const int kp = std::min(KP, K - i0);       // passes i0 .. i0 + kp - 1 as one batch of kp * B images
        const int R = kp * M;                      // rows of this batch
        const size_t RF = (size_t)R * F;
        const int32_t* tgt = tgt_all + (size_t)i0 * M;
        auto site = [&](int k) { return (unsigned)(8 * i0 + k); };      // of the batch's first pass; pass p draws site + 8 p
        // ---- forward: model.decode (model.py:86-103) — the embeddings and the queries are dropped afresh in every pass -----------
        CHK(dropout_add(cx, content0, true, nullptr, content, ME, kp, drop, site(S_CONTENT)));
        CHK((run_layernorm<float>(s, content, P(p + "norm_c.weight"), P(p + "norm_c.bias"), cn, nullptr, R, E, eps)));
        CHK(lin_fwd(cx, cn, sa_w + (size_t)E * E, sa_b + E, nullptr, 0, kvc, R, 2 * E, E));
        hipLaunchKernelGGL(dropout_rows_passes_kernel, dim3((unsigned)((ME + 255) / 256), (unsigned)kp), dim3(256), 0, s, pq, L, E, qd, ME, drop, site(S_QUERY));
        HIPCHK(hipGetLastError());
        CHK((run_layernorm<float>(s, qd, P(p + "norm_q.weight"), P(p + "norm_q.bias"), qn, nullptr, R, E, eps)));
        CHK(lin_fwd(cx, qn, sa_w, sa_b, nullptr, 0, qsa, R, E, E));
        // ---- DecoderLayer.forward_stream (modules.py:55-79), Decoder.norm (:124), head (model.py:63) -----------------------------
        sa.qmask = query_masks + (size_t)i0 * L * L; sa.drop_site = site(S_SA_PROB);
        CHK(train_attn(cx, sa, kp * B, false, 32));
        CHK(lin_fwd(cx, sa_o, P(p + "self_attn.out_proj.weight"), P(p + "self_attn.out_proj.bias"), nullptr, 0, pm, R, E, E));
        CHK(dropout_add(cx, pm, false, qd, t1, ME, kp, drop, site(S_SA_OUT)));
        CHK((run_layernorm<float>(s, t1, P(p + "norm1.weight"), P(p + "norm1.bias"), n1, nullptr, R, E, eps)));
        CHK(lin_fwd(cx, n1, ca_w, ca_b, nullptr, 0, q2, R, E, E));
        ca.drop_site = site(S_CA_PROB);
        if (ca_loop) { ca.pass_loop = kp; ca.kv_accumulate = i0 > 0 ? 1 : 0; }
        CHK(train_attn(cx, ca, kp * B, false, 32));
        CHK(lin_fwd(cx, ca_o, P(p + "cross_attn.out_proj.weight"), P(p + "cross_attn.out_proj.bias"), nullptr, 0, pm, R, E, E));
        CHK(dropout_add(cx, pm, false, t1, t2, ME, kp, drop, site(S_CA_OUT)));
        CHK((run_layernorm<float>(s, t2, P(p + "norm2.weight"), P(p + "norm2.bias"), n2, nullptr, R, E, eps)));
        CHK(lin_fwd(cx, n2, P(p + "linear1.weight"), P(p + "linear1.bias"), nullptr, 0, hpre, R, F, E, hact));      // hact = gelu(hpre): the product's epilogue (bf16-operand mode) or gelu_fwd_kernel
        if (drop.thresh) CHK(dropout_add(cx, hact, false, nullptr, hact, MF, kp, drop, site(S_FF_HIDDEN)));
        CHK(lin_fwd(cx, hact, P(p + "linear2.weight"), P(p + "linear2.bias"), nullptr, 0, pm, R, E, F));
        CHK(dropout_add(cx, pm, false, t2, t3, ME, kp, drop, site(S_FF_OUT)));
        CHK((run_layernorm<float>(s, t3, P("decoder.norm.weight"), P("decoder.norm.bias"), out, nullptr, R, E, eps)));
        CHK(lin_fwd(cx, out, P("head.weight"), P("head.bias"), nullptr, 0, logits, R, C, E));
        hipLaunchKernelGGL(ce_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, s, logits, tgt, R, C, m->cfg.pad_id, row_loss);
        HIPCHK(hipGetLastError());
        for (int q = 0; q < kp; ++q) {             // each pass's own mean (system.py:189-190), rows summed in the order its own launch would
            hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss + (size_t)q * M, tgt + (size_t)q * M, M, m->cfg.pad_id, losses + i0 + q,
                               counts + i0 + q);
            HIPCHK(hipGetLastError());
        }
        // ---- backward ------------------------------------------------------------------------------------------------------------
        hipLaunchKernelGGL(ce_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, s, logits, tgt, R, C, m->cfg.pad_id, 1.0f / (float)total_targets);
        HIPCHK(hipGetLastError());
        CHK(lin_bwd(cx, out, P("head.weight"), logits, G("head.weight"), G("head.bias"), d_a, R, C, E));                                    // d_a = d out
        CHK(ln_bwd(cx, t3, P("decoder.norm.weight"), d_a, nullptr, d_b, G("decoder.norm.weight"), G("decoder.norm.bias"), nullptr, R, E, eps));  // d_b = d t3
        CHK(dropout_add(cx, d_b, false, nullptr, pm, ME, kp, drop, site(S_FF_OUT)));
        CHK(lin_bwd(cx, hact, P(p + "linear2.weight"), pm, G(p + "linear2.weight"), G(p + "linear2.bias"), d_h, R, E, F));                  // d_h = d hact
        if (drop.thresh && MF % 4 == 0)            // d_h = d hpre: the MLP's inner dropout and the GELU backward in one pass
            hipLaunchKernelGGL(gelu_bwd_drop_passes_kernel, dim3((unsigned)((MF + 1023) / 1024), (unsigned)kp), dim3(256), 0, s, hpre, d_h, d_h, MF, drop, site(S_FF_HIDDEN));
        else {
            if (drop.thresh) CHK(dropout_add(cx, d_h, false, nullptr, d_h, MF, kp, drop, site(S_FF_HIDDEN)));
            hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((RF + 1023) / 1024)), dim3(256), 0, s, hpre, d_h, d_h, RF);
        }
        HIPCHK(hipGetLastError());
        CHK(lin_bwd(cx, n2, P(p + "linear1.weight"), d_h, G(p + "linear1.weight"), G(p + "linear1.bias"), d_a, R, F, E));                   // d_a = d n2
        CHK(ln_bwd(cx, t2, P(p + "norm2.weight"), d_a, d_b, d_b, G(p + "norm2.weight"), G(p + "norm2.bias"), nullptr, R, E, eps));          // d_b = d t2
        CHK(dropout_add(cx, d_b, false, nullptr, pm, ME, kp, drop, site(S_CA_OUT)));
        CHK(lin_bwd(cx, ca_o, P(p + "cross_attn.out_proj.weight"), pm, G(p + "cross_attn.out_proj.weight"), G(p + "cross_attn.out_proj.bias"),
                    d_c, R, E, E));                                                                                                        // d_c = d ca_o
        CHK(train_attn(cx, ca, kp * B, true, 32));                                                                                          // d_a = d q2; d_kvm_p[pass] = (KP == 1: d_kvm +=)
        if (KP > 1 && !ca_loop) {                  // d_kvm (+)= the batch's passes, in ascending order
            hipLaunchKernelGGL(sum_passes_kernel, dim3((unsigned)(((size_t)MS * 2 * E / 4 + 255) / 256)), dim3(256), 0, s, d_kvm_p, d_kvm, (size_t)MS * 2 * E, kp,
                               i0 > 0 ? 1 : 0);
            HIPCHK(hipGetLastError());
        }
        CHK(lin_bwd(cx, n1, ca_w, d_a, G(p + "cross_attn.in_proj_weight"), G(p + "cross_attn.in_proj_bias"), d_c, R, E, E));                // d_c = d n1
        CHK(ln_bwd(cx, t1, P(p + "norm1.weight"), d_c, d_b, d_a, G(p + "norm1.weight"), G(p + "norm1.bias"), nullptr, R, E, eps));          // d_a = d t1
        CHK(dropout_add(cx, d_a, false, nullptr, pm, ME, kp, drop, site(S_SA_OUT)));
        CHK(lin_bwd(cx, sa_o, P(p + "self_attn.out_proj.weight"), pm, G(p + "self_attn.out_proj.weight"), G(p + "self_attn.out_proj.bias"),
                    d_b, R, E, E));                                                                                                        // d_b = d sa_o
        CHK(train_attn(cx, sa, kp * B, true, 32));                                                                                          // d_qb = d q; d_kvc =
        CHK(lin_bwd(cx, qn, sa_w, d_qb, G(p + "self_attn.in_proj_weight"), G(p + "self_attn.in_proj_bias"), d_c, R, E, E));                 // d_c = d qn
        CHK(ln_bwd(cx, qd, P(p + "norm_q.weight"), d_c, d_a, d_b, G(p + "norm_q.weight"), G(p + "norm_q.bias"), nullptr, R, E, eps));       // d_b = d qd
        CHK(dropout_add(cx, d_b, false, nullptr, d_b, ME, kp, drop, site(S_QUERY)));
        CHK(colsum(cx, d_b, (long)L * E, kp * B, L * E, d_pq, true));                           // every image's query rows are pos_queries[l]
        CHK(lin_bwd(cx, cn, sa_w + (size_t)E * E, d_kvc, G(p + "self_attn.in_proj_weight") + (size_t)E * E, G(p + "self_attn.in_proj_bias") + E,
                    d_c, R, 2 * E, E));                                                                                                    // d_c = d cn
        CHK(ln_bwd(cx, content, P(p + "norm_c.weight"), d_c, nullptr, d_b, G(p + "norm_c.weight"), G(p + "norm_c.bias"), nullptr, R, E, eps));  // d_b = d content
        // d_content (+)= every pass's d content through that pass's mask (the passes in ascending order)
        hipLaunchKernelGGL(dropout_sum_passes_kernel, dim3((unsigned)((ME + 255) / 256)), dim3(256), 0, s, d_b, d_content, ME, kp, drop, site(S_CONTENT), i0 > 0 ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, s, losses, counts, K, losses + K);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(loss_out, losses + K, sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(loss_out + 1, losses, (size_t)K * sizeof(float), hipMemcpyDeviceToDevice, s));

    // ---- what every permutation shares, once ---------------------------------------------------------------------------------------
    if (L > 1) CHK(colsum(cx, d_content + E, (long)L * E, B, (L - 1) * E, d_pq, true));       // content row j carries pos_queries[j - 1]
    {   // token-embedding gradient: the B * L rows in chunks of 768, one workgroup per (token id, chunk), then the chunks folded in order
        const int rows_per = 768, chunks = (M + rows_per - 1) / rows_per;
        if (chunks > 1 && (size_t)m->cfg.num_tokens * chunks * E <= cx.scratch_floats) {
            hipLaunchKernelGGL(embed_bwd_kernel, dim3(m->cfg.num_tokens, chunks), dim3(256), 0, s, d_content, tokens, L, B, L, E, sqrtE,
                               G("text_embed.embedding.weight"), cx.scratch, rows_per);
            hipLaunchKernelGGL(embed_bwd_fold_kernel, dim3(m->cfg.num_tokens), dim3(256), 0, s, cx.scratch, chunks, E, sqrtE, G("text_embed.embedding.weight"));
        } else {
            hipLaunchKernelGGL(embed_bwd_kernel, dim3(m->cfg.num_tokens, 1), dim3(256), 0, s, d_content, tokens, L, B, L, E, sqrtE,
                               G("text_embed.embedding.weight"), (float*)nullptr, M);
        }
        HIPCHK(hipGetLastError());
    }
    CHK(lin_bwd(cx, memory, ca_w + (size_t)E * E, d_kvm, G(p + "cross_attn.in_proj_weight") + (size_t)E * E, G(p + "cross_attn.in_proj_bias") + E,
                dmemory, MS, 2 * E, E));
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)(((size_t)L * E + 255) / 256)), dim3(256), 0, s, G("pos_queries"), d_pq, G("pos_queries"), (size_t)L * E);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- training step, encoder side: forward that keeps what the backward needs, and the backward ------------------------------
struct TrainEncoderLayout {          // offsets in floats
    size_t patches, layer0, layer_stride, x_last, n, hact, d_x, d_a, d_h, dqkv, tmp, scratch, scratch2, scratch_floats, total;      // scratch2: the backward's second stream
    size_t w16, w16_layer, d_x16, d_h16;      // bf16 shadows (train_enc_shadows): the Linear weights and their transposes ([layer][qkv, proj, fc1, fc2][W16 | Wt16]),
                                              // the residual-stream gradient and the fc1-output gradient
    size_t x(int i) const { return layer0 + i * layer_stride; }
    size_t qkv, ao, x_mid, hpre, hact_l, n1, n2;     // offsets inside one layer's record (x at 0); hact_l, n1, n2: the GELU output and the two
                                             // LayerNorm outputs, kept for the backward (round 3: they used to be recomputed there — a 600 MB and two
                                             // 150 MB passes per block; the record grows from 10 E to 16 E floats per token per block)
};
static TrainEncoderLayout train_encoder_layout(const parseq_model* m, int B) {
    const size_t E = m->cfg.embed_dim, F = E * m->cfg.enc_mlp_ratio, MS = (size_t)B * m->tokens, PK = m->patch_k;
    TrainEncoderLayout o;
    size_t off = 0;
    auto take = [&](size_t n) { const size_t at = off; off += (n + 63) / 64 * 64; return at; };
    o.patches = take(MS * PK);
    o.layer0 = off;
    take(MS * E); o.qkv = off - o.layer0; take(MS * 3 * E); o.ao = off - o.layer0; take(MS * E); o.x_mid = off - o.layer0; take(MS * E);
    o.hpre = off - o.layer0; take(MS * F);
    o.hact_l = off - o.layer0; take(MS * F);
    o.n1 = off - o.layer0; take(MS * E); o.n2 = off - o.layer0; take(MS * E);
    o.layer_stride = off - o.layer0;
    off = o.layer0 + o.layer_stride * (size_t)m->cfg.enc_depth;
    o.x_last = take(MS * E); o.n = take(MS * E); o.hact = take(MS * F); o.d_x = take(MS * E); o.d_a = take(MS * E); o.d_h = take(MS * F);
    o.dqkv = take(MS * 3 * E); o.tmp = take(MS * E);
    o.scratch_floats = train_scratch_floats(MS, E); o.scratch = take(o.scratch_floats);
    // the second scratch belongs to the backward's second stream, which only the bf16-operand mode has (the fp32 mode carves nothing for it)
    o.scratch2 = m->train_precision == PARSEQ_BF16 ? take(o.scratch_floats) : o.scratch;
    o.w16_layer = 4 * E * E + 2 * E * F;      // floats = 2 bf16 each: W16 and Wt16 of the block's four Linear weights
    o.w16 = take(o.w16_layer * (size_t)m->cfg.enc_depth); o.d_x16 = take(MS * E / 2 + 8); o.d_h16 = take(MS * F / 2 + 8);
    o.total = off;
    return o;
}

extern "C" size_t parseq_train_encoder_workspace_bytes(const parseq_model* m, int batch) {
    if (!m || batch <= 0) return 0;
    return train_encoder_layout(m, batch).total * sizeof(float);
}

static int train_encoder_check(const parseq_model* m, int batch, const void* workspace, size_t workspace_bytes) {
    if (!m || !workspace) return fail(PARSEQ_E_INVALID, "null argument");
    if (m->vitstr) return fail(PARSEQ_E_INVALID, "the training step is built for PARSeq only");
    if (batch <= 0) return fail(PARSEQ_E_INVALID, "batch %d", batch);
    for (const ParamSpec& ps : m->params) if (!ps.set) return fail(PARSEQ_E_STATE, "parameter %s was never set", ps.key.c_str());
    const size_t need = train_encoder_layout(m, batch).total * sizeof(float);
    if (workspace_bytes < need) return fail(PARSEQ_E_INVALID, "workspace: %zu bytes given, %zu needed", workspace_bytes, need);
    return 0;
}

template <typename TO>
static int train_ln_fwd(hipStream_t s, const float* x, const float* w, const float* b, TO* out, int rows, int E, float eps) {
    if (E > 768 || E % 2) return fail(PARSEQ_E_INVALID, "training layernorm: E=%d", E);
    hipLaunchKernelGGL((ln_fwd_kernel<TO>), dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, out, rows, E, eps);
    HIPCHK(hipGetLastError());
    return 0;
}
// bf16 shadow operands for the encoder's products (train_ops.h SgemmArgs): the bf16-operand mode at the shapes the bf16 attention kernel
// and the 64-deep GEMM take.  In that mode the record's n1 / n2 / ao / hact_l slots hold bf16 (in the first half of the fp32 slot).
// PARSEQ_TRAIN_NO_SHADOWS=1 keeps every operand fp32 in memory (the A/B and the bit-identity test).
static bool train_enc_shadows(const parseq_model* m) {
    const int E = m->cfg.embed_dim, F = E * m->cfg.enc_mlp_ratio;
    return m->train_precision == PARSEQ_BF16 && E % 64 == 0 && F % 64 == 0 && m->tokens == TB_N && E == m->cfg.enc_heads * TB_HD &&
           !getenv("PARSEQ_TRAIN_F32_ATTN") && !getenv("PARSEQ_TRAIN_NO_SHADOWS");
}
// Level 2 (the default with shadows on): tensors that exist ONLY to be rounded to bf16 by their consumers or to feed a GELU derivative are
// stored as bf16 and nothing else — the fc1 pre-activation (its GELU derivative is taken at the bf16 value), the gradient of the fc1
// output and the gradient of q | k | v (the dW products read them as bf16 too; the bias gradients are sums of the bf16 values) — and the
// fc2 / proj dW products read the residual-stream gradient through its bf16 shadow (the fp32 copy stays: LayerNorm backward adds to it).  That is
// what bf16-mixed autocast keeps of these tensors (BASELINE configs[4]); it is no longer bit-identical to the fp32-in-memory path — the
// oracle gates of the bf16-operand mode hold it.  PARSEQ_TRAIN_SHADOW_LEVEL=1: shadows beside the fp32 copies only (bit-identical).
static bool train_enc_bf16_only(const parseq_model* m) {
    const char* lv = getenv("PARSEQ_TRAIN_SHADOW_LEVEL");
    return train_enc_shadows(m) && !(lv && lv[0] == '1');
}
// The forward as one launch: the bf16-only record (level 2) at the geometry encoder_blocks.h is written for (E = 384, six 64-wide heads,
// 128 tokens, hidden = 4 E) with a block's record inside one 32-bit buffer descriptor.  PARSEQ_TRAIN_ENC_PER_OP=1: the per-operation
// launches (the A/B; also what every other geometry and record mode runs).
static bool train_enc_one_launch(const parseq_model* m, const TrainEncoderLayout& o) {
    return train_enc_bf16_only(m) && m->cfg.embed_dim == 384 && m->cfg.enc_mlp_ratio == 4 && m->tokens == 128 &&
           o.layer_stride * sizeof(float) < ((size_t)1 << 32) && !getenv("PARSEQ_TRAIN_ENC_PER_OP");
}
struct EncShadowW { bf16_t* w; bf16_t* wt; };
// which: 0 attn.qkv [3E, E], 1 attn.proj [E, E], 2 mlp.fc1 [F, E], 3 mlp.fc2 [E, F]
static EncShadowW enc_shadow_w(const TrainEncoderLayout& o, float* ws, int layer, int which, size_t E, size_t F) {
    bf16_t* base = reinterpret_cast<bf16_t*>(ws + o.w16 + o.w16_layer * (size_t)layer);
    const size_t at[4] = {0, 6 * E * E, 8 * E * E, 8 * E * E + 2 * E * F}, n[4] = {3 * E * E, E * E, E * F, E * F};
    return EncShadowW{base + at[which], base + at[which] + n[which]};
}

static TrainAttnArgs enc_attn_args(const parseq_model* m, float* qkv, float* ao, const float* d_ao, float* dqkv) {
    const int E = m->cfg.embed_dim, S = m->tokens;
    TrainAttnArgs a{};
    a.q = qkv; a.q_bstride = (long)S * 3 * E; a.ldq = 3 * E; a.k = qkv + E; a.v = qkv + 2 * E; a.ldkv = 3 * E;
    a.o = ao; a.ldo = E; a.d_o = d_ao; a.dq = dqkv; a.lddq = 3 * E; a.dk = dqkv ? dqkv + E : nullptr; a.dv = dqkv ? dqkv + 2 * E : nullptr;
    a.lddkv = 3 * E; a.kv_accumulate = 0; a.Lq = S; a.Lk = S; a.H = m->cfg.enc_heads; a.scale = 1.0f / sqrtf((float)ATT_HD);
    return a;
}

extern "C" int parseq_train_encoder_forward(parseq_model* m, const float* images, int batch, float* memory_out, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    if (!images || !memory_out) return fail(PARSEQ_E_INVALID, "null argument");
    CHK(train_encoder_check(m, batch, workspace, workspace_bytes));
    DevGuard dg(m->device);
    const TrainEncoderLayout o = train_encoder_layout(m, batch);
    hipStream_t s = (hipStream_t)stream;
    const int E = m->cfg.embed_dim, F = E * m->cfg.enc_mlp_ratio, S = m->tokens, MS = batch * S, PK = m->patch_k;
    const float eps = m->cfg.enc_ln_eps;
    float* w = reinterpret_cast<float*>(workspace);
    auto P = [&](const std::string& key) { return m->p(m->enc + key); };
    const TrainCtx cx{s, w + o.scratch, m->train_precision == PARSEQ_BF16, o.scratch_floats};
    hipLaunchKernelGGL(patches_kernel, dim3(MS), dim3(256), 0, s, images, m->cfg.img_h, m->cfg.img_w, m->cfg.patch_h, m->cfg.patch_w, w + o.patches);
    HIPCHK(hipGetLastError());
    CHK(lin_fwd(cx, w + o.patches, P("patch_embed.proj.weight"), P("patch_embed.proj.bias"), P("pos_embed"), S, w + o.x(0), MS, E, PK));
    const size_t elems = (size_t)MS * F;
    const bool shadows = train_enc_shadows(m), only16 = train_enc_bf16_only(m);
    m->enc_record_mode = (shadows ? 1 : 0) | (only16 ? 2 : 0);      // what the record's slots hold; the backward entry must read them the same way
    m->enc_record_ws = workspace;
    if (shadows) {
        // this step's weights as bf16, both ways round (the backward entry reads the transposes from the same workspace): one launch over
        // the 4 * depth matrices (it was one launch per matrix: 48 launches of 2 - 7 us); the table is a model constant
        if (!m->shadow_tab_dev) {
            const char* names[4] = {"attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"};
            const int wn[4] = {3 * E, E, F, E}, wk[4] = {E, E, E, F};
            std::vector<ShadowEntry> tab;
            unsigned tile = 0;
            for (int i = 0; i < m->cfg.enc_depth; ++i)
                for (int j = 0; j < 4; ++j) {
                    const EncShadowW sw = enc_shadow_w(o, w, i, j, E, F);
                    tab.push_back(ShadowEntry{(unsigned)m->params[m->index.at(m->enc + "blocks." + std::to_string(i) + "." + names[j])].offset, (unsigned)wn[j], (unsigned)wk[j],
                                              tile, (unsigned long long)(sw.w - reinterpret_cast<bf16_t*>(w + o.w16))});
                    tile += (unsigned)((wn[j] / 32) * (wk[j] / 32));
                }
            // published only when valid: a failed upload must not leave a non-null table of garbage offsets behind for the next call
            void* dev = nullptr;
            HIPCHK(hipMalloc(&dev, tab.size() * sizeof(ShadowEntry)));
            if (hipMemcpy(dev, tab.data(), tab.size() * sizeof(ShadowEntry), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(dev);
                return fail(PARSEQ_E_HIP, "upload of the weight-shadow table failed: %s", hipGetErrorString(hipGetLastError()));
            }
            m->shadow_tiles = (int)tile;
            m->shadow_tab_dev = dev;
        }
        hipLaunchKernelGGL(weight_shadows_kernel, dim3(m->shadow_tiles), dim3(256), 0, s, m->master, reinterpret_cast<const ShadowEntry*>(m->shadow_tab_dev),
                           4 * m->cfg.enc_depth, reinterpret_cast<bf16_t*>(w + o.w16));
        HIPCHK(hipGetLastError());
    }
    if (train_enc_one_launch(m, o)) {
        // The twelve blocks as ONE launch (encoder_blocks.h record mode): the inference throughput kernel's walk — x resident in the
        // accumulators, the weights streamed from this step's bf16 shadows — writing the record on the way; 48 GEMM, 24 LayerNorm and
        // 12 attention launches and every re-read of an activation disappear, what is left is the record's own bytes.
        const int depth = m->cfg.enc_depth;
        if (!m->train_blocks_dev) {      // a model constant: parameter offsets into the master, weight offsets relative to the shadows' base
            std::vector<EncBlockParams> tab(depth);
            auto off = [&](const std::string& key) { return (unsigned)m->params[m->index.at(m->enc + key)].offset; };
            for (int i = 0; i < depth; ++i) {
                const std::string b = "blocks." + std::to_string(i) + ".";
                const unsigned w0 = (unsigned)(2 * o.w16_layer * (size_t)i), EE = (unsigned)(E * E), EF = (unsigned)(E * F);      // bf16 elements from the shadows' base
                EncBlockParams& e = tab[i];
                e.ln1_w = off(b + "norm1.weight"); e.ln1_b = off(b + "norm1.bias"); e.bqkv = off(b + "attn.qkv.bias"); e.bproj = off(b + "attn.proj.bias");
                e.ln2_w = off(b + "norm2.weight"); e.ln2_b = off(b + "norm2.bias"); e.b1 = off(b + "mlp.fc1.bias"); e.b2 = off(b + "mlp.fc2.bias");
                e.wqkv = w0; e.wproj = w0 + 6 * EE; e.w1 = w0 + 8 * EE; e.w2 = w0 + 8 * EE + 2 * EF;       // enc_shadow_w's W16 of each pair
            }
            EncBlockParams* dev = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&dev), depth * sizeof(EncBlockParams)));
            if (hipMemcpy(dev, tab.data(), depth * sizeof(EncBlockParams), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(dev);
                return fail(PARSEQ_E_HIP, "upload of the training block table failed: %s", hipGetErrorString(hipGetLastError()));
            }
            m->train_blocks_dev = dev;
        }
        const EncBlockParams* tab = m->train_blocks_dev;
        const EncRecordParams rec{w + o.layer0, o.layer_stride, (unsigned)(o.layer_stride * sizeof(float)), (unsigned)(o.qkv * 4), (unsigned)(o.ao * 4),
                                  (unsigned)(o.x_mid * 4), (unsigned)(o.hpre * 4), (unsigned)(o.hact_l * 4), (unsigned)(o.n1 * 4), (unsigned)(o.n2 * 4)};
        HIPCHK((launch_enc_blocks_record<384>(s, w + o.x_last, reinterpret_cast<const bf16_t*>(w + o.w16), o.w16_layer * (size_t)depth * sizeof(float),
                                              m->master, tab, depth, eps, MS, rec)));
        return run_layernorm<float>(s, w + o.x_last, P("norm.weight"), P("norm.bias"), memory_out, nullptr, MS, E, eps);
    }
    for (int i = 0; i < m->cfg.enc_depth; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        float* x = w + o.x(i); float* qkv = x + o.qkv; float* ao = x + o.ao; float* x_mid = x + o.x_mid; float* hpre = x + o.hpre;
        float* x_out = i + 1 < m->cfg.enc_depth ? w + o.x(i + 1) : w + o.x_last;
        if (shadows) {
            bf16_t* n1 = reinterpret_cast<bf16_t*>(x + o.n1); bf16_t* n2 = reinterpret_cast<bf16_t*>(x + o.n2);
            bf16_t* ao16 = reinterpret_cast<bf16_t*>(ao); bf16_t* hact16 = reinterpret_cast<bf16_t*>(x + o.hact_l);
            CHK(train_ln_fwd(s, x, P(p + "norm1.weight"), P(p + "norm1.bias"), n1, MS, E, eps));
            CHK(lin_fwd16(cx, n1, enc_shadow_w(o, w, i, 0, E, F).w, P(p + "attn.qkv.bias"), nullptr, 0, qkv, MS, 3 * E, E));
            TrainAttnArgs aa = enc_attn_args(m, qkv, ao, nullptr, nullptr);
            aa.o16 = ao16;
            CHK(train_attn(cx, aa, batch, false, ATT_HD));
            CHK(lin_fwd16(cx, ao16, enc_shadow_w(o, w, i, 1, E, F).w, P(p + "attn.proj.bias"), x, MS, x_mid, MS, E, E));
            CHK(train_ln_fwd(s, x_mid, P(p + "norm2.weight"), P(p + "norm2.bias"), n2, MS, E, eps));
            if (only16) CHK(lin_fwd16(cx, n2, enc_shadow_w(o, w, i, 2, E, F).w, P(p + "mlp.fc1.bias"), nullptr, 0, nullptr, MS, F, E, hact16, reinterpret_cast<bf16_t*>(hpre)));
            else CHK(lin_fwd16(cx, n2, enc_shadow_w(o, w, i, 2, E, F).w, P(p + "mlp.fc1.bias"), nullptr, 0, hpre, MS, F, E, hact16));
            CHK(lin_fwd16(cx, hact16, enc_shadow_w(o, w, i, 3, E, F).w, P(p + "mlp.fc2.bias"), x_mid, MS, x_out, MS, E, F));
            continue;
        }
        CHK(train_ln_fwd(s, x, P(p + "norm1.weight"), P(p + "norm1.bias"), x + o.n1, MS, E, eps));
        CHK(lin_fwd(cx, x + o.n1, P(p + "attn.qkv.weight"), P(p + "attn.qkv.bias"), nullptr, 0, qkv, MS, 3 * E, E));
        CHK(train_attn(cx, enc_attn_args(m, qkv, ao, nullptr, nullptr), batch, false, ATT_HD));
        CHK(lin_fwd(cx, ao, P(p + "attn.proj.weight"), P(p + "attn.proj.bias"), x, MS, x_mid, MS, E, E));
        CHK(train_ln_fwd(s, x_mid, P(p + "norm2.weight"), P(p + "norm2.bias"), x + o.n2, MS, E, eps));
        float* hact_l = x + o.hact_l;
        CHK(lin_fwd(cx, x + o.n2, P(p + "mlp.fc1.weight"), P(p + "mlp.fc1.bias"), nullptr, 0, hpre, MS, F, E, hact_l));      // hpre and gelu(hpre), one epilogue
        CHK(lin_fwd(cx, hact_l, P(p + "mlp.fc2.weight"), P(p + "mlp.fc2.bias"), x_mid, MS, x_out, MS, E, F));
    }
    return run_layernorm<float>(s, w + o.x_last, P("norm.weight"), P("norm.bias"), memory_out, nullptr, MS, E, eps);
}

// ---- gradient segments: the hook for overlapping the data-parallel all-reduce with the encoder's backward -----------------------------
// The reference trains under Lightning's DDP strategy (train.py:65-71): gradient buckets are all-reduced while the backward is still
// running.  Here the gradients live in ONE flat buffer laid out like the master weights — pos_queries | encoder.pos_embed, patch_embed |
// blocks 0 .. depth-1 | encoder.norm | decoder.*, head.*, text_embed.* — and become final in this order: the decoder's part (and
// pos_queries) before the encoder's backward starts, then encoder.norm and the blocks from the last to the first, then pos_embed /
// patch_embed.  Segment k (completion order) is a contiguous range of the buffer; parseq_train_encoder_backward records event k on its
// stream right after the last kernel that writes into it:
//   0: [decoder begin, end)            1: [block depth-1 begin, decoder begin)  (with encoder.norm)
//   1 + j: block depth-1-j, j = 1 .. depth-2          depth: [0, block 0 end)   (pos_queries, pos_embed, patch_embed, block 0)
static int64_t grad_block_begin(const parseq_model* m, int i) { return (int64_t)m->params[m->index.at(m->enc + "blocks." + std::to_string(i) + ".norm1.weight")].offset; }
static int grad_segment_range(const parseq_model* m, int k, int64_t* begin, int64_t* end) {
    const int depth = m->cfg.enc_depth;
    if (m->vitstr || depth < 2) return fail(PARSEQ_E_INVALID, "gradient segments are defined for the PARSeq training step (depth >= 2)");
    if (k < 0 || k > depth) return fail(PARSEQ_E_INVALID, "gradient segment %d outside [0, %d]", k, depth);
    const int64_t dec = (int64_t)m->params[m->index.at("decoder.layers.0.self_attn.in_proj_weight")].offset;
    if (k == 0) { *begin = dec; *end = (int64_t)m->master_elems; }
    else if (k == 1) { *begin = grad_block_begin(m, depth - 1); *end = dec; }
    else if (k < depth) { *begin = grad_block_begin(m, depth - k); *end = grad_block_begin(m, depth - k + 1); }
    else { *begin = 0; *end = grad_block_begin(m, 1); }
    return 0;
}
static int grad_event_record(parseq_model* m, int k, hipStream_t s) {
    while ((int)m->grad_events.size() <= m->cfg.enc_depth) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->grad_events.push_back(e);
    }
    HIPCHK(hipEventRecord(m->grad_events[k], s));
    return 0;
}
extern "C" int parseq_train_grad_segments(const parseq_model* m) { return (m && !m->vitstr && m->cfg.enc_depth >= 2) ? m->cfg.enc_depth + 1 : 0; }
extern "C" int parseq_train_grad_segment(parseq_model* m, int index, int64_t* begin, int64_t* end, void** event) {
    if (!m || !begin || !end) return fail(PARSEQ_E_INVALID, "null argument");
    CHK(grad_segment_range(m, index, begin, end));
    if (event) {
        if ((int)m->grad_events.size() <= index || !m->grad_events_valid)
            return fail(PARSEQ_E_STATE, "gradient segment %d: its event does not belong to this step (no parseq_train_encoder_backward has run to its end since the last parseq_train_decoder)", index);
        *event = (void*)m->grad_events[index];
    }
    return 0;
}
extern "C" int parseq_stream_wait_event(void* stream, void* event) {
    if (!event) return fail(PARSEQ_E_INVALID, "null event");
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return 0;
}

extern "C" int parseq_train_encoder_backward(parseq_model* m, const float* dmemory, int batch, float* grads, void* workspace, size_t workspace_bytes,
                                             void* stream) {
    if (!dmemory || !grads) return fail(PARSEQ_E_INVALID, "null argument");
    CHK(train_encoder_check(m, batch, workspace, workspace_bytes));
    DevGuard dg(m->device);
    const TrainEncoderLayout o = train_encoder_layout(m, batch);
    hipStream_t s = (hipStream_t)stream;
    const int E = m->cfg.embed_dim, F = E * m->cfg.enc_mlp_ratio, S = m->tokens, MS = batch * S, PK = m->patch_k;
    const float eps = m->cfg.enc_ln_eps;
    float* w = reinterpret_cast<float*>(workspace);
    auto P = [&](const std::string& key) { return m->p(m->enc + key); };
    auto G = [&](const std::string& key) { return grads + m->params[m->index.at(m->enc + key)].offset; };
    float* d_x = w + o.d_x; float* d_a = w + o.d_a; float* d_h = w + o.d_h; float* dqkv = w + o.dqkv;
    float* tmp = w + o.tmp;
    const size_t elems = (size_t)MS * F;
    const TrainCtx cx{s, w + o.scratch, m->train_precision == PARSEQ_BF16, o.scratch_floats};
    const bool shadows = train_enc_shadows(m), only16 = train_enc_bf16_only(m);
    if (m->enc_record_ws == workspace && m->enc_record_mode != ((shadows ? 1 : 0) | (only16 ? 2 : 0)))
        return fail(PARSEQ_E_STATE, "training encoder backward: the record in this workspace was written in mode %d, this call would read it in mode %d "
                    "(PARSEQ_TRAIN_NO_SHADOWS / PARSEQ_TRAIN_SHADOW_LEVEL / train precision changed between forward and backward)", m->enc_record_mode,
                    (shadows ? 1 : 0) | (only16 ? 2 : 0));
    bf16_t* d_x16 = shadows ? reinterpret_cast<bf16_t*>(w + o.d_x16) : nullptr;
    bf16_t* d_h16 = shadows ? reinterpret_cast<bf16_t*>(w + o.d_h16) : nullptr;
    const bool segs = m->cfg.enc_depth >= 2;
    const bool two_streams = only16 && !getenv("PARSEQ_TRAIN_ONE_STREAM");
    hipStream_t side = nullptr;
    hipEvent_t* side_ev = nullptr;
    if (two_streams) {
        // the caller stream's own side stream and events (a step split into micro-batches runs several backwards at once, each on its own stream)
        size_t slot = 0;
        while (slot < m->train_sides.size() && m->train_sides[slot].key != s) ++slot;
        if (slot == m->train_sides.size()) {
            if (slot >= 8) return fail(PARSEQ_E_STATE, "training encoder backward: more than 8 distinct caller streams on one model");
            parseq_model::TrainSide ts;
            ts.key = s;
            HIPCHK(hipStreamCreateWithFlags(&ts.side, hipStreamNonBlocking));
            for (hipEvent_t& e : ts.ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            m->train_sides.push_back(ts);
        }
        side = m->train_sides[slot].side;
        side_ev = m->train_sides[slot].ev;
    }
    const TrainCtx cxs{side, w + o.scratch2, m->train_precision == PARSEQ_BF16, o.scratch_floats};
    m->grad_events_valid = false;
    if (segs) CHK(grad_event_record(m, 0, s));      // the decoder's gradients were written by parseq_train_decoder, earlier on this stream
    CHK(ln_bwd(cx, w + o.x_last, P("norm.weight"), dmemory, nullptr, d_x, G("norm.weight"), G("norm.bias"), tmp, MS, E, eps, d_x16));
    for (int i = m->cfg.enc_depth - 1; i >= 0; --i) {
        // block i + 1 (and, behind the last block, encoder.norm) is final: segment depth - 1 - i
        if (segs && i < m->cfg.enc_depth - 1 && i >= 0) CHK(grad_event_record(m, m->cfg.enc_depth - 1 - i, s));
        const std::string p = "blocks." + std::to_string(i) + ".";
        float* x = w + o.x(i); float* qkv = x + o.qkv; float* ao = x + o.ao; float* x_mid = x + o.x_mid; float* hpre = x + o.hpre;
        if (shadows) {
            const bf16_t* n1 = reinterpret_cast<const bf16_t*>(x + o.n1); const bf16_t* n2 = reinterpret_cast<const bf16_t*>(x + o.n2);
            const bf16_t* ao16 = reinterpret_cast<const bf16_t*>(ao); const bf16_t* hact16 = reinterpret_cast<const bf16_t*>(x + o.hact_l);
            if (two_streams) {
                // The block's four weight-gradient products on the SECOND stream, beside the chain dX -> LayerNorm / attention backward -> dX that
                // needs them for nothing: each starts when its dY exists (an event of the main stream) and is waited for only where the main
                // stream is about to overwrite that dY (the residual-stream gradient's shadow d_x16) or closes the block.  Same kernels, same
                // operands, same order inside every buffer: bit-identical to the one-stream schedule (PARSEQ_TRAIN_ONE_STREAM=1).
                // (an error return inside the block must not leave weight-gradient kernels running on the hidden stream behind the caller's back:
                // the block is a lambda and a failure joins the side stream before it is reported)
                auto block = [&]() -> int {
                hipEvent_t* ev = side_ev;
                const bf16_t* hpre16 = reinterpret_cast<const bf16_t*>(hpre);
                bf16_t* dqkv16 = reinterpret_cast<bf16_t*>(dqkv);
                HIPCHK(hipEventRecord(ev[0], s)); HIPCHK(hipStreamWaitEvent(side, ev[0], 0));                    // d_x16 of this block exists
                CHK(lin_bwd16_dw(cxs, hact16, nullptr, d_x16, G(p + "mlp.fc2.weight"), G(p + "mlp.fc2.bias"), MS, E, F));
                HIPCHK(hipEventRecord(ev[4], side));
                CHK(lin_bwd16_dx(cx, enc_shadow_w(o, w, i, 3, E, F).wt, nullptr, d_x16, nullptr, d_h16, MS, E, F, nullptr, hpre16));
                HIPCHK(hipEventRecord(ev[1], s)); HIPCHK(hipStreamWaitEvent(side, ev[1], 0));                    // d_h16 exists
                CHK(lin_bwd16_dw(cxs, n2, nullptr, d_h16, G(p + "mlp.fc1.weight"), G(p + "mlp.fc1.bias"), MS, F, E));
                CHK(lin_bwd16_dx(cx, enc_shadow_w(o, w, i, 2, E, F).wt, nullptr, d_h16, d_a, nullptr, MS, F, E));
                HIPCHK(hipStreamWaitEvent(s, ev[4], 0));                                                          // fc2's dW has read d_x16
                CHK(ln_bwd(cx, x_mid, P(p + "norm2.weight"), d_a, d_x, d_x, G(p + "norm2.weight"), G(p + "norm2.bias"), tmp, MS, E, eps, d_x16));
                HIPCHK(hipEventRecord(ev[2], s)); HIPCHK(hipStreamWaitEvent(side, ev[2], 0));                    // the new d_x16 exists
                CHK(lin_bwd16_dw(cxs, ao16, nullptr, d_x16, G(p + "attn.proj.weight"), G(p + "attn.proj.bias"), MS, E, E));
                HIPCHK(hipEventRecord(ev[5], side));
                CHK(lin_bwd16_dx(cx, enc_shadow_w(o, w, i, 1, E, F).wt, nullptr, d_x16, d_a, nullptr, MS, E, E));
                TrainAttnArgs ab = enc_attn_args(m, qkv, ao, d_a, dqkv);
                ab.dq16 = dqkv16; ab.dk16 = dqkv16 + E; ab.dv16 = dqkv16 + 2 * E;
                CHK(train_attn(cx, ab, batch, true, ATT_HD));
                HIPCHK(hipEventRecord(ev[3], s)); HIPCHK(hipStreamWaitEvent(side, ev[3], 0));                    // dqkv16 exists
                CHK(lin_bwd16_dw(cxs, n1, nullptr, dqkv16, G(p + "attn.qkv.weight"), G(p + "attn.qkv.bias"), MS, 3 * E, E));
                HIPCHK(hipEventRecord(ev[6], side));
                CHK(lin_bwd16_dx(cx, enc_shadow_w(o, w, i, 0, E, F).wt, nullptr, dqkv16, d_a, nullptr, MS, 3 * E, E));
                HIPCHK(hipStreamWaitEvent(s, ev[5], 0));                                                          // proj's dW has read d_x16
                CHK(ln_bwd(cx, x, P(p + "norm1.weight"), d_a, d_x, d_x, G(p + "norm1.weight"), G(p + "norm1.bias"), tmp, MS, E, eps, d_x16));
                HIPCHK(hipStreamWaitEvent(s, ev[6], 0));       // the block's gradients are final on the main stream too (the segment event that follows covers them)
                return 0;
                };
                const int rc = block();
                if (rc) {
                    (void)hipStreamSynchronize(side);          // whatever was enqueued there has finished with `grads` and the workspace before the caller hears of the failure
                    return rc;
                }
                continue;
            }
            if (only16) {
                CHK(lin_bwd16(cx, hact16, enc_shadow_w(o, w, i, 3, E, F).wt, nullptr, d_x16, G(p + "mlp.fc2.weight"), G(p + "mlp.fc2.bias"), nullptr, d_h16, MS, E, F,
                              nullptr, reinterpret_cast<const bf16_t*>(hpre)));
                CHK(lin_bwd16(cx, n2, enc_shadow_w(o, w, i, 2, E, F).wt, nullptr, d_h16, G(p + "mlp.fc1.weight"), G(p + "mlp.fc1.bias"), d_a, nullptr, MS, F, E));
            } else {
                CHK(lin_bwd16(cx, hact16, enc_shadow_w(o, w, i, 3, E, F).wt, d_x, d_x16, G(p + "mlp.fc2.weight"), G(p + "mlp.fc2.bias"), d_h, d_h16, MS, E, F, hpre));
                CHK(lin_bwd16(cx, n2, enc_shadow_w(o, w, i, 2, E, F).wt, d_h, d_h16, G(p + "mlp.fc1.weight"), G(p + "mlp.fc1.bias"), d_a, nullptr, MS, F, E));
            }
            CHK(ln_bwd(cx, x_mid, P(p + "norm2.weight"), d_a, d_x, d_x, G(p + "norm2.weight"), G(p + "norm2.bias"), tmp, MS, E, eps, d_x16));
            CHK(lin_bwd16(cx, ao16, enc_shadow_w(o, w, i, 1, E, F).wt, only16 ? nullptr : d_x, d_x16, G(p + "attn.proj.weight"), G(p + "attn.proj.bias"), d_a, nullptr, MS, E, E));
            if (only16) {
                TrainAttnArgs ab = enc_attn_args(m, qkv, ao, d_a, dqkv);
                bf16_t* dqkv16 = reinterpret_cast<bf16_t*>(dqkv);
                ab.dq16 = dqkv16; ab.dk16 = dqkv16 + E; ab.dv16 = dqkv16 + 2 * E;
                CHK(train_attn(cx, ab, batch, true, ATT_HD));
                CHK(lin_bwd16(cx, n1, enc_shadow_w(o, w, i, 0, E, F).wt, nullptr, dqkv16, G(p + "attn.qkv.weight"), G(p + "attn.qkv.bias"), d_a, nullptr, MS, 3 * E, E));
            } else {
                CHK(train_attn(cx, enc_attn_args(m, qkv, ao, d_a, dqkv), batch, true, ATT_HD));
                CHK(lin_bwd16(cx, n1, enc_shadow_w(o, w, i, 0, E, F).wt, dqkv, nullptr, G(p + "attn.qkv.weight"), G(p + "attn.qkv.bias"), d_a, nullptr, MS, 3 * E, E));
            }
            CHK(ln_bwd(cx, x, P(p + "norm1.weight"), d_a, d_x, d_x, G(p + "norm1.weight"), G(p + "norm1.bias"), tmp, MS, E, eps, d_x16));
            continue;
        }
        // x_out = x_mid + fc2(gelu(fc1(norm2(x_mid))))        d_x = d x_out
        const float* hact = x + o.hact_l;                        // kept by the forward
        CHK(lin_bwd(cx, hact, P(p + "mlp.fc2.weight"), d_x, G(p + "mlp.fc2.weight"), G(p + "mlp.fc2.bias"), d_h, MS, E, F, hpre));      // d_h = d hpre (GELU backward folded in)
        CHK(lin_bwd(cx, x + o.n2, P(p + "mlp.fc1.weight"), d_h, G(p + "mlp.fc1.weight"), G(p + "mlp.fc1.bias"), d_a, MS, F, E));
        CHK(ln_bwd(cx, x_mid, P(p + "norm2.weight"), d_a, d_x, d_x, G(p + "norm2.weight"), G(p + "norm2.bias"), tmp, MS, E, eps));   // d_x = d x_mid
        // x_mid = x + proj(attention(qkv(norm1(x))))
        CHK(lin_bwd(cx, ao, P(p + "attn.proj.weight"), d_x, G(p + "attn.proj.weight"), G(p + "attn.proj.bias"), d_a, MS, E, E));        // d_a = d ao
        CHK(train_attn(cx, enc_attn_args(m, qkv, ao, d_a, dqkv), batch, true, ATT_HD));
        CHK(lin_bwd(cx, x + o.n1, P(p + "attn.qkv.weight"), dqkv, G(p + "attn.qkv.weight"), G(p + "attn.qkv.bias"), d_a, MS, 3 * E, E));
        CHK(ln_bwd(cx, x, P(p + "norm1.weight"), d_a, d_x, d_x, G(p + "norm1.weight"), G(p + "norm1.bias"), tmp, MS, E, eps));          // d_x = d x
    }
    CHK(colsum(cx, d_x, (long)S * E, batch, S * E, G("pos_embed"), true));
    CHK(lin_bwd(cx, w + o.patches, P("patch_embed.proj.weight"), d_x, G("patch_embed.proj.weight"), G("patch_embed.proj.bias"), nullptr, MS, E, PK));
    if (segs) CHK(grad_event_record(m, m->cfg.enc_depth, s));      // block 0, patch_embed, pos_embed (and pos_queries): everything is final
    m->grad_events_valid = segs;
    return 0;
}

// ---- training step, optimiser ---------------------------------------------------------------------------------------------
extern "C" int parseq_grad_norm(const float* grads, int64_t n, float* norm_out, float* workspace, void* stream) {
    if (!grads || !norm_out || !workspace || n <= 0) return fail(PARSEQ_E_INVALID, "null / empty argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, grads, (size_t)n, workspace);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, workspace, norm_out);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int parseq_adamw_step(parseq_model* m, const float* grads, float* exp_avg, float* exp_avg_sq, const int32_t* decay_flags, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_norm, float max_norm,
                                 void* stream) {
    if (!m || !grads || !exp_avg || !exp_avg_sq) return fail(PARSEQ_E_INVALID, "null argument");
    if (step < 1) return fail(PARSEQ_E_INVALID, "step %d: steps count from 1", step);
    DevGuard dg(m->device);
    for (const ParamSpec& ps : m->params) if (!ps.set) return fail(PARSEQ_E_STATE, "parameter %s was never set", ps.key.c_str());
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    // runs of consecutive tensors with the same weight-decay flag are one launch (with weight_decay == 0, the reference's
    // configuration, the whole buffer is)
    const int np = (int)m->params.size();
    int i = 0;
    while (i < np) {
        const bool decay = decay_flags && weight_decay != 0.f && decay_flags[i];
        int j = i + 1;
        while (j < np && (decay_flags && weight_decay != 0.f && decay_flags[j]) == decay) ++j;
        const size_t lo = m->params[i].offset, hi = j < np ? m->params[j].offset : m->master_elems;
        hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, s, m->master + lo, grads + lo, exp_avg + lo,
                           exp_avg_sq + lo, hi - lo, lr, beta1, beta2, eps, decay ? weight_decay : 0.f, bc1, bc2_sqrt, grad_norm, max_norm);
        HIPCHK(hipGetLastError());
        i = j;
    }
    m->version++;
    return 0;
}

extern "C" int parseq_model_get_param(const parseq_model* m, const char* key, float* device_ptr, int64_t numel, void* stream) {
    if (!m || !key || !device_ptr) return fail(PARSEQ_E_INVALID, "null argument");
    auto it = m->index.find(key);
    if (it == m->index.end()) return fail(PARSEQ_E_INVALID, "unknown parameter key '%s'", key);
    const ParamSpec& ps = m->params[it->second];
    if (ps.numel != numel) return fail(PARSEQ_E_INVALID, "parameter %s: numel %lld, expected %lld", key, (long long)numel, (long long)ps.numel);
    DevGuard dg(m->device);
    HIPCHK(hipMemcpyAsync(device_ptr, m->master + ps.offset, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// Every parameter out in ONE launch (the per-tensor form above is ~170 small device copies per optimiser step: 0.6 ms of a 34 ms step).
// The table of copy pieces is rebuilt only when the destination pointers change (never, in a training loop).
extern "C" int parseq_model_get_params(parseq_model* m, float* const* device_ptrs, int count, void* stream) {
    if (!m || !device_ptrs) return fail(PARSEQ_E_INVALID, "null argument");
    if (count != (int)m->params.size()) return fail(PARSEQ_E_INVALID, "%d destination pointers for %d parameters", count, (int)m->params.size());
    for (int i = 0; i < count; ++i) if (!device_ptrs[i]) return fail(PARSEQ_E_INVALID, "parameter %s: null destination", m->params[i].key.c_str());
    DevGuard dg(m->device);
    hipStream_t s = (hipStream_t)stream;
    if (!m->out_chunks || m->out_ptrs.size() != (size_t)count || !std::equal(m->out_ptrs.begin(), m->out_ptrs.end(), device_ptrs)) {
        std::vector<CopyPiece> pieces;
        for (int i = 0; i < count; ++i) {
            const ParamSpec& ps = m->params[i];
            for (int64_t at = 0; at < ps.numel; at += COPY_PIECE_ELEMS)
                pieces.push_back(CopyPiece{m->master + ps.offset + at, device_ptrs[i] + at, (int)std::min<int64_t>(COPY_PIECE_ELEMS, ps.numel - at)});
        }
        HIPCHK(hipStreamSynchronize(s));      // a launch that still reads the old table
        if (m->out_chunks) { (void)hipFree(m->out_chunks); m->out_chunks = nullptr; }
        HIPCHK(hipMalloc(&m->out_chunks, pieces.size() * sizeof(CopyPiece)));
        HIPCHK(hipMemcpy(m->out_chunks, pieces.data(), pieces.size() * sizeof(CopyPiece), hipMemcpyHostToDevice));
        m->out_chunk_count = (int)pieces.size();
        m->out_ptrs.assign(device_ptrs, device_ptrs + count);
    }
    hipLaunchKernelGGL(copy_pieces_kernel, dim3((unsigned)m->out_chunk_count), dim3(256), 0, s, reinterpret_cast<const CopyPiece*>(m->out_chunks));
    HIPCHK(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------------------------------
// input resize
