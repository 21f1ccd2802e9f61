// libparseq_hip.so — input resize, post-process, and the per-kernel entry points the parity tests call.
#include "lib_internal.h"

// -------------------------------------------------------------------------------------------------------------------
static int resize_taps(int in_size, int out_size) {
    const double scale = (double)in_size / (double)out_size;
    const double support = 2.0 * (scale < 1.0 ? 1.0 : scale);
    return (int)ceil(support) * 2 + 1;
}

extern "C" size_t parseq_resize_workspace_bytes(int batch) { return batch > 0 ? (size_t)batch * sizeof(ImageDesc) : 0; }

extern "C" int parseq_resize_bicubic(const parseq_image_desc* images, int batch, int out_h, int out_w, uint8_t* out, void* workspace,
                                     void* stream) {
    static_assert(sizeof(parseq_image_desc) == sizeof(ImageDesc), "descriptor layouts must match");
    if (!images || !out || !workspace) return fail(PARSEQ_E_INVALID, "null images / out / workspace");
    if (batch <= 0 || out_h <= 0 || out_w <= 0) return fail(PARSEQ_E_INVALID, "bad shape: batch %d, output %dx%d", batch, out_h, out_w);
    int ksh = 1, ksv = 1;
    for (int i = 0; i < batch; ++i) {
        const parseq_image_desc& d = images[i];
        if (!d.data || d.height <= 0 || d.width <= 0 || d.row_stride < (int64_t)d.width * 3)
            return fail(PARSEQ_E_INVALID, "image %d: bad descriptor (%dx%d, row stride %lld)", i, d.height, d.width, (long long)d.row_stride);
        ksh = std::max(ksh, resize_taps(d.width, out_w));
        ksv = std::max(ksv, resize_taps(d.height, out_h));
    }
    const size_t lds = sizeof(int) * ((size_t)out_w * ksh + (size_t)out_h * ksv + 2 * (size_t)(out_w + out_h));
    if (lds > 150 * 1024) return fail(PARSEQ_E_INVALID, "an image is too large for the on-chip weight tables (%zu bytes of LDS needed)", lds);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(workspace, images, (size_t)batch * sizeof(ImageDesc), hipMemcpyHostToDevice, s));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_bicubic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(resize_bicubic_kernel, dim3(batch), dim3(256), lds, s, reinterpret_cast<const ImageDesc*>(workspace), out_h, out_w, ksh, ksv, out);
    HIPCHK(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------------------------------
// post-processing
// -------------------------------------------------------------------------------------------------------------------
extern "C" int parseq_postprocess(const float* logits, int batch, int L, int C, int eos_id, int32_t* ids_out, int32_t* lengths_out,
                                  float* probs_out, float* confidence_out, void* stream) {
    if (!logits || !ids_out || !lengths_out) return fail(PARSEQ_E_INVALID, "null logits / ids_out / lengths_out");
    if (batch <= 0 || L < 1 || L > 64 || C < 1) return fail(PARSEQ_E_INVALID, "bad shape: batch %d, L %d (1..64), C %d", batch, L, C);
    if (eos_id < 0 || eos_id >= C) return fail(PARSEQ_E_INVALID, "eos_id %d outside [0, %d)", eos_id, C);
    hipLaunchKernelGGL(postprocess_kernel, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, batch, L, C, eos_id, ids_out,
                       lengths_out, probs_out, confidence_out);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int parseq_cross_entropy(const float* logits, const int32_t* targets, int rows, int C, int ignore_index, float* loss_out,
                                    int32_t* numel_out, float* workspace, void* stream) {
    if (!logits || !targets || !loss_out || !numel_out || !workspace) return fail(PARSEQ_E_INVALID, "null argument");
    if (rows <= 0 || C <= 0) return fail(PARSEQ_E_INVALID, "bad shape: rows %d, C %d", rows, C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, targets, rows, C, ignore_index, workspace);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, s, workspace, targets, rows, ignore_index, loss_out, numel_out);
    HIPCHK(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------------------------------
// single operators
// -------------------------------------------------------------------------------------------------------------------
extern "C" int parseq_op_layernorm(const float* x, const float* w, const float* b, void* y, int out_dtype, int rows, int E, float eps, void* stream) {
    CHK(check_arch());
    if (!x || !w || !b || !y || rows <= 0) return fail(PARSEQ_E_INVALID, "bad argument");
    if (out_dtype == PARSEQ_BF16) return run_layernorm<bf16_t>((hipStream_t)stream, x, w, b, (bf16_t*)y, nullptr, rows, E, eps);
    return run_layernorm<float>((hipStream_t)stream, x, w, b, (float*)y, nullptr, rows, E, eps);
}

template <typename T>
static int op_linear_impl(const T* A, const T* W, const float* bias, void* C, int act, int M, int N, int K, hipStream_t s) {
    if (act) return run_gemm<T>(s, ARowMajor<T>{A, K}, W, K, M, N, K, epi_gelu<T>(M, N, bias, (T*)C, N));
    return run_gemm<T>(s, ARowMajor<T>{A, K}, W, K, M, N, K, epi_store<float>(M, N, bias, (float*)C, N));
}

extern "C" int parseq_op_linear(const void* A, const void* W, const float* bias, void* C, int dtype, int act, int M, int N, int K, void* stream) {
    CHK(check_arch());
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || (K % 8)) return fail(PARSEQ_E_INVALID, "bad argument (K must be a multiple of 8)");
    if (act && (N % 4)) return fail(PARSEQ_E_INVALID, "act=1 needs N %% 4 == 0");
    if (dtype == PARSEQ_BF16) return op_linear_impl<bf16_t>((const bf16_t*)A, (const bf16_t*)W, bias, C, act, M, N, K, (hipStream_t)stream);
    SplitScope ss(dtype == PARSEQ_BF16X3);          // A: f32; W: the block-planar hi / lo copy made by parseq_op_split_pack
    return op_linear_impl<float>((const float*)A, (const float*)W, bias, C, act, M, N, K, (hipStream_t)stream);
}

// C[M, N] (fp32) = LayerNorm(x[M, 384]; gamma, beta, eps) W^T + bias through the generic tile GEMM with the LayerNorm fused into the
// A-operand loader (the decoder's q-projection / linear1 / head form); dtype PARSEQ_F32 or PARSEQ_BF16X3 (W then block-planar).
extern "C" int parseq_op_ln_linear(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, float* C_,
                                   int dtype, int M, int N, float eps, void* stream) {
    CHK(check_arch());
    if (!x || !gamma || !beta || !W || !C_ || M <= 0 || N <= 0) return fail(PARSEQ_E_INVALID, "bad argument");
    if (dtype != PARSEQ_F32 && dtype != PARSEQ_BF16X3) return fail(PARSEQ_E_INVALID, "dtype %d", dtype);
    SplitScope ss(dtype == PARSEQ_BF16X3);
    return run_gemm<float>((hipStream_t)stream, ALayerNorm<float, 384>{x, gamma, beta, eps, 0, nullptr}, (const float*)W, 384, M, N, 384,
                           epi_store<float>(M, N, bias, C_, N));
}

extern "C" int parseq_op_ln_linear_pairs(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, void* C_, void* ws,
                                         int act, int M, int N, float eps, void* stream) {
    CHK(check_arch());
    if (!x || !gamma || !beta || !W || !C_ || !ws || M <= 0 || N <= 0) return fail(PARSEQ_E_INVALID, "bad argument");
    if (act && (N % 32)) return fail(PARSEQ_E_INVALID, "pair-layout output: N=%d is not a multiple of 32", N);
    hipStream_t s = (hipStream_t)stream;
    constexpr int E = 384;
    CHK((run_layernorm_split(s, x, gamma, beta, reinterpret_cast<unsigned char*>(ws), M, E, eps)));
    const bf16_t* A2 = reinterpret_cast<const bf16_t*>(ws);
    const bf16_t* W2 = reinterpret_cast<const bf16_t*>(W);
    if (act) {
        EpiGeluSplit eg; static_cast<EpiBase&>(eg) = epi_base(M, N, bias); eg.out = reinterpret_cast<unsigned char*>(C_); eg.ldo = N;
        HIPCHK((launch_gemm_pairs<128, 128, 2, 2>(s, A2, 2 * E, W2, 2 * E, M, N, 2 * E, eg)));
    } else {
        HIPCHK((launch_gemm_pairs<128, 128, 2, 2>(s, A2, 2 * E, W2, 2 * E, M, N, 2 * E, epi_store<float>(M, N, bias, (float*)C_, N))));
    }
    return 0;
}

extern "C" int parseq_op_split_pack(const float* src, void* dst, int64_t numel, void* stream) {
    CHK(check_arch());
    if (!src || !dst || numel <= 0 || (numel % 32)) return fail(PARSEQ_E_INVALID, "bad argument (numel must be a multiple of 32)");
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((numel / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (unsigned char*)dst, (size_t)numel);
    HIPCHK(hipGetLastError());
    return 0;
}

// Tile-configuration sweep hook for tools/gemm_bench.py (not used by the product path, which picks via run_gemm).
template <typename T>
static int op_linear_cfg_impl(const T* A, const T* W, const float* bias, void* C, int act, int M, int N, int K, int cfg, hipStream_t s) {
#define PQ_CFG(ID, BM, BN, WM, WN, KB, NBUF)                                                                                     \
    case 100 + ID: {                                                                                                              \
        EpiNull en; static_cast<EpiBase&>(en) = epi_base(M, N, bias); en.sink = (float*)C;                                        \
        HIPCHK((launch_gemm<T, BM, BN, WM, WN, KB, NBUF>(s, ARowMajor<T>{A, K}, W, K, M, N, K, en)));                              \
        return 0; }                                                                                                               \
    case ID:                                                                                                                      \
        if (act) HIPCHK((launch_gemm<T, BM, BN, WM, WN, KB, NBUF>(s, ARowMajor<T>{A, K}, W, K, M, N, K, epi_gelu<T>(M, N, bias, (T*)C, N)))); \
        else HIPCHK((launch_gemm<T, BM, BN, WM, WN, KB, NBUF>(s, ARowMajor<T>{A, K}, W, K, M, N, K, epi_store<float>(M, N, bias, (float*)C, N)))); \
        return 0;
    switch (cfg) {
        PQ_CFG(0, 128, 128, 2, 2, 128, 2)
        PQ_CFG(1, 128, 128, 2, 2, 256, 1)
        PQ_CFG(2, 256, 128, 4, 2, 128, 2)
        PQ_CFG(3, 256, 128, 4, 2, 256, 1)
        PQ_CFG(4, 128, 128, 2, 2, 256, 2)
        PQ_CFG(5, 64, 64, 2, 2, 768, 1)
        PQ_CFG(6, 128, 128, 2, 2, 128, 1)
        PQ_CFG(7, 256, 128, 4, 2, 128, 1)
        default: return fail(PARSEQ_E_INVALID, "unknown gemm cfg %d", cfg);
    }
#undef PQ_CFG
}

extern "C" int parseq_op_linear_cfg(const void* A, const void* W, const float* bias, void* C, int dtype, int act, int M, int N, int K, int cfg, void* stream) {
    CHK(check_arch());
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || (K % 8) || (act && (N % 4))) return fail(PARSEQ_E_INVALID, "bad argument");
    if (dtype == PARSEQ_BF16) return op_linear_cfg_impl<bf16_t>((const bf16_t*)A, (const bf16_t*)W, bias, C, act, M, N, K, cfg, (hipStream_t)stream);
    return op_linear_cfg_impl<float>((const float*)A, (const float*)W, bias, C, act, M, N, K, cfg, (hipStream_t)stream);
}

// x += fc2(gelu(fc1(LayerNorm(x)))) through the fused MLP kernel (E = 384, hidden 1536, bf16 weights).
extern "C" int parseq_op_mlp(float* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                             const float* b2, int M, void* stream) {
    CHK(check_arch());
    if (!x || !gamma || !beta || !W1 || !b1 || !W2 || !b2 || M <= 0) return fail(PARSEQ_E_INVALID, "bad argument");
    HIPCHK((launch_fused_mlp<384>((hipStream_t)stream, x, gamma, beta, 1e-6f, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, M)));
    return 0;
}

extern "C" int parseq_op_mlp_variant(float* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                                     const float* b2, int M, int variant, void* stream) {
    CHK(check_arch());
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 0: HIPCHK((launch_fused_mlp<384>(s, x, gamma, beta, 1e-6f, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, M))); break;             // x re-read by the epilogue
        case 10: HIPCHK((launch_fused_mlp<384, true>(s, x, gamma, beta, 1e-6f, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, M))); break;      // x resident in the accumulators
        case 11: HIPCHK((launch_mlp_branch<384>(s, x, gamma, beta, 1e-6f, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, M))); break;           // shared-phase form (encoder_blocks.h)
        default: return fail(PARSEQ_E_INVALID, "variant %d", variant);
    }
    return 0;
}

extern "C" int parseq_op_attn_fused(float* x, const float* gamma, const float* beta, const void* Wqkv, const float* bqkv, const void* Wproj,
                                    const float* bproj, int M, int variant, void* stream) {
    CHK(check_arch());
    if (!x || !gamma || !beta || !Wqkv || !bqkv || !Wproj || !bproj || M <= 0 || (M % 128)) return fail(PARSEQ_E_INVALID, "bad argument (M must be a multiple of 128: whole images)");
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 0: HIPCHK((launch_fused_attn<384>(s, x, gamma, beta, 1e-6f, (const bf16_t*)Wqkv, bqkv, (const bf16_t*)Wproj, bproj, M))); break;
        case 1: HIPCHK((launch_attn_branch<384>(s, x, gamma, beta, 1e-6f, (const bf16_t*)Wqkv, bqkv, (const bf16_t*)Wproj, bproj, M))); break;   // shared-phase form (encoder_blocks.h)
        default: return fail(PARSEQ_E_INVALID, "variant %d", variant);
    }
    return 0;
}

extern "C" int parseq_op_enc_blocks(float* x, const void* const* block_ptrs, int depth, int M, void* table_ws, void* stream) {
    CHK(check_arch());
    if (!x || !block_ptrs || !table_ws || depth <= 0 || M <= 0 || (M % 128)) return fail(PARSEQ_E_INVALID, "bad argument (M must be a multiple of 128: whole images)");
    // the kernel addresses the bf16 matrices relative to one base through a 32-bit buffer descriptor and the fp32 vectors relative to
    // another: take the lowest address of each kind as the base
    static const int kW[4] = {2, 4, 8, 10};                                  // wqkv, wproj, w1, w2
    static const size_t kWElems[4] = {(size_t)1152 * 384, (size_t)384 * 384, (size_t)1536 * 384, (size_t)384 * 1536};
    uintptr_t wlo = ~(uintptr_t)0, whi = 0, plo = ~(uintptr_t)0, phi = 0;
    for (int i = 0; i < depth; ++i) {
        const void* const* q = block_ptrs + (size_t)i * 12;
        for (int k = 0; k < 12; ++k) {
            if (!q[k]) return fail(PARSEQ_E_INVALID, "block %d: null pointer %d", i, k);
            const uintptr_t a = reinterpret_cast<uintptr_t>(q[k]);
            int wi = -1;
            for (int j = 0; j < 4; ++j) if (kW[j] == k) wi = j;
            if (wi >= 0) { wlo = std::min(wlo, a); whi = std::max(whi, a + kWElems[wi] * 2); }
            else { plo = std::min(plo, a); phi = std::max(phi, a + 1536 * 4); }
        }
    }
    if (whi - wlo >= ((uintptr_t)1 << 32) || phi - plo >= ((uintptr_t)1 << 34) || (wlo & 1) || (plo & 3))
        return fail(PARSEQ_E_INVALID, "block parameters are spread over more than 4 GiB of address space");
    std::vector<EncBlockParams> host(depth);
    for (int i = 0; i < depth; ++i) {
        const void* const* q = block_ptrs + (size_t)i * 12;
        unsigned o[12];
        for (int k = 0; k < 12; ++k) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(q[k]);
            const bool isw = k == 2 || k == 4 || k == 8 || k == 10;
            if ((a - (isw ? wlo : plo)) % (isw ? 2 : 4)) return fail(PARSEQ_E_INVALID, "block %d: pointer %d is misaligned", i, k);
            o[k] = (unsigned)((a - (isw ? wlo : plo)) / (isw ? 2 : 4));
        }
        EncBlockParams& e = host[i];
        e.ln1_w = o[0]; e.ln1_b = o[1]; e.wqkv = o[2]; e.bqkv = o[3]; e.wproj = o[4]; e.bproj = o[5];
        e.ln2_w = o[6]; e.ln2_b = o[7]; e.w1 = o[8]; e.b1 = o[9]; e.w2 = o[10]; e.b2 = o[11];
    }
    HIPCHK(hipMemcpy(table_ws, host.data(), host.size() * sizeof(EncBlockParams), hipMemcpyHostToDevice));      // test hook: synchronous upload
    HIPCHK((launch_enc_blocks<384>((hipStream_t)stream, x, reinterpret_cast<const bf16_t*>(wlo), (size_t)(whi - wlo), reinterpret_cast<const float*>(plo),
                                   reinterpret_cast<const EncBlockParams*>(table_ws), depth, 1e-6f, M)));
    return 0;
}

// The head and the tail of the one-launch encoder with no blocks in between (encoder_blocks.h patch_head / kv_phase), for sharp
// per-kernel tests.  images != NULL: x = patches(images) Wpe^T + posb is computed in the accumulators (else x is loaded);
// kmem != NULL: the launch ends with K | V = LayerNorm(x; norm_w, norm_b) Wkv^T + bkv as bf16 head-split rows (else x is stored).
extern "C" int parseq_op_enc_head_tail(float* x, const void* images, int images_dtype, const void* wpe, const float* posb,
                                       const float* norm_w, const float* norm_b, const void* wkv, const float* bkv, void* kmem, void* vmem,
                                       int M, void* stream) {
    CHK(check_arch());
    if (M <= 0 || (M % 128)) return fail(PARSEQ_E_INVALID, "M must be a multiple of 128 (whole images)");
    const bool head = images != nullptr, tail = kmem != nullptr;
    if (!head && !tail) return fail(PARSEQ_E_INVALID, "neither images (head) nor kmem (tail) given");
    if (head && (!wpe || !posb || (images_dtype != PARSEQ_F32 && images_dtype != PARSEQ_BF16 && images_dtype != PARSEQ_U8)))
        return fail(PARSEQ_E_INVALID, "head: wpe, posb and an image dtype of f32 / bf16 / u8 are required");
    if (tail && (!vmem || !norm_w || !norm_b || !wkv || !bkv)) return fail(PARSEQ_E_INVALID, "tail: vmem, norm_w, norm_b, wkv, bkv are required");
    if ((!head || !tail) && !x) return fail(PARSEQ_E_INVALID, "x is required unless both head and tail are given");
    uintptr_t wlo = ~(uintptr_t)0, whi = 0, plo = ~(uintptr_t)0;
    auto span_w = [&](const void* q, size_t elems) { const uintptr_t a = reinterpret_cast<uintptr_t>(q); wlo = std::min(wlo, a); whi = std::max(whi, a + elems * 2); };
    if (head) span_w(wpe, (size_t)384 * 96);
    if (tail) span_w(wkv, (size_t)768 * 384);
    // the head's DMA pieces put "row offset - LDS immediate" into the scalar offset (StreamLane::issue_v): with the 192-byte rows of Wpe
    // that is negative for a weight at the very start of the descriptor (EB_HEAD_MIN_WPE).  The product's pack has pos_embed ahead of
    // it; here the descriptor simply starts 4 KiB below the lowest weight (addresses below it are never formed)
    if (wlo >= 4096) wlo -= 4096;
    if (tail) for (const float* q : {norm_w, norm_b, bkv}) plo = std::min(plo, reinterpret_cast<uintptr_t>(q));
    if (whi - wlo >= ((uintptr_t)1 << 32) || (wlo & 1)) return fail(PARSEQ_E_INVALID, "weights are spread over more than 4 GiB of address space");
    EncTailParams et{0, 0, 0, 0, nullptr, nullptr, 12};
    if (tail) {
        auto poff = [&](const float* q) { return (unsigned)((reinterpret_cast<uintptr_t>(q) - plo) / 4); };
        for (const float* q : {norm_w, norm_b, bkv})
            if ((reinterpret_cast<uintptr_t>(q) - plo) >= ((uintptr_t)1 << 34) || ((reinterpret_cast<uintptr_t>(q) - plo) & 3)) return fail(PARSEQ_E_INVALID, "tail vectors are spread too far apart");
        et.norm_w = poff(norm_w); et.norm_b = poff(norm_b); et.bkv = poff(bkv);
        et.wkv = (unsigned)((reinterpret_cast<uintptr_t>(wkv) - wlo) / 2);
        et.kmem = reinterpret_cast<bf16_t*>(kmem); et.vmem = reinterpret_cast<bf16_t*>(vmem);
    }
    EncHeadParams eh{nullptr, 0, 0, nullptr};
    if (head) {
        eh.images = images;
        eh.img_dtype = images_dtype == PARSEQ_U8 ? EB_IMG_U8 : (images_dtype == PARSEQ_BF16 ? EB_IMG_BF16 : EB_IMG_F32);
        eh.wpe = (unsigned)((reinterpret_cast<uintptr_t>(wpe) - wlo) / 2); eh.posb = posb;
    }
    HIPCHK((launch_enc_blocks<384>((hipStream_t)stream, x, reinterpret_cast<const bf16_t*>(wlo), (size_t)(whi - wlo),
                                   reinterpret_cast<const float*>(tail ? plo : reinterpret_cast<uintptr_t>(posb)), nullptr, 0, 1e-6f, M, et, eh)));
    return 0;
}

// `depth` encoder blocks in one launch in the bf16x3 arithmetic (encoder_blocks_x3.h).  `master`: ONE f32 buffer holding every parameter
// of the blocks (each tensor on a 32-element boundary); `pack`: its block-planar hi | lo copy (parseq_op_split_pack over the whole
// buffer); offsets: HOST array of depth * 12 element offsets into `master`, per block in EncBlockParams order (norm1 w, b, Wqkv, bqkv,
// Wproj, bproj, norm2 w, b, W1, b1, W2, b2).  tail_offsets (HOST, 4 element offsets: final norm w, b, Wkv [768, 384], bkv [768]) with
// kmem / vmem (f32 [M / 128][12][128][32]) != NULL: instead of storing x the launch ends with K | V = LayerNorm(x) Wkv^T + bkv.
static int enc_blocks_x3_op(bool eight_waves, float* x, const float* master, const void* pack, int64_t master_elems, const uint32_t* offsets, int depth,
                            int M, void* table_ws, float* scratch, const uint32_t* tail_offsets, float* kmem, float* vmem, void* stream) {
    CHK(check_arch());
    if (!x || !master || !pack || !offsets || !table_ws || !scratch || depth <= 0 || M <= 0 || (M % 128) || master_elems <= 0 || (master_elems % 32))
        return fail(PARSEQ_E_INVALID, "bad argument (M must be a multiple of 128: whole images; master_elems a multiple of 32)");
    if ((kmem || vmem) && (!kmem || !vmem || !tail_offsets)) return fail(PARSEQ_E_INVALID, "tail: kmem, vmem and tail_offsets go together");
    std::vector<EncBlockParams> host(depth);
    for (int i = 0; i < depth; ++i) {
        const uint32_t* o = offsets + (size_t)i * 12;
        for (int k = 0; k < 12; ++k) if (o[k] % 32 || (int64_t)o[k] >= master_elems) return fail(PARSEQ_E_INVALID, "block %d: offset %d (%u) is not a 32-element boundary inside the buffer", i, k, o[k]);
        EncBlockParams& e = host[i];
        e.ln1_w = o[0]; e.ln1_b = o[1]; e.wqkv = o[2]; e.bqkv = o[3]; e.wproj = o[4]; e.bproj = o[5];
        e.ln2_w = o[6]; e.ln2_b = o[7]; e.w1 = o[8]; e.b1 = o[9]; e.w2 = o[10]; e.b2 = o[11];
    }
    HIPCHK(hipMemcpy(table_ws, host.data(), host.size() * sizeof(EncBlockParams), hipMemcpyHostToDevice));      // test hook: synchronous upload
    x3::EncTailX3 et{0, 0, 0, 0, nullptr, nullptr, 12};
    if (kmem) { et.norm_w = tail_offsets[0]; et.norm_b = tail_offsets[1]; et.wkv = tail_offsets[2]; et.bkv = tail_offsets[3]; et.kmem = kmem; et.vmem = vmem; }
    if (eight_waves)
        HIPCHK((x3w::launch_enc_blocks_x3w<384>((hipStream_t)stream, x, pack, (size_t)master_elems * sizeof(float), master,
                                                reinterpret_cast<const EncBlockParams*>(table_ws), depth, 1e-6f, M, scratch, et)));
    else
        HIPCHK((x3::launch_enc_blocks_x3<384>((hipStream_t)stream, x, pack, (size_t)master_elems * sizeof(float), master,
                                              reinterpret_cast<const EncBlockParams*>(table_ws), depth, 1e-6f, M, scratch, et)));
    return 0;
}
extern "C" int parseq_op_enc_blocks_x3(float* x, const float* master, const void* pack, int64_t master_elems, const uint32_t* offsets, int depth,
                                       int M, void* table_ws, float* scratch, const uint32_t* tail_offsets, float* kmem, float* vmem, void* stream) {
    return enc_blocks_x3_op(false, x, master, pack, master_elems, offsets, depth, M, table_ws, scratch, tail_offsets, kmem, vmem, stream);
}
// the same through the two-waves-per-SIMD kernel (encoder_blocks_x3w.h: what parseq_forward runs); bit-identical to parseq_op_enc_blocks_x3
extern "C" int parseq_op_enc_blocks_x3w(float* x, const float* master, const void* pack, int64_t master_elems, const uint32_t* offsets, int depth,
                                        int M, void* table_ws, float* scratch, const uint32_t* tail_offsets, float* kmem, float* vmem, void* stream) {
    return enc_blocks_x3_op(true, x, master, pack, master_elems, offsets, depth, M, table_ws, scratch, tail_offsets, kmem, vmem, stream);
}

// LayerNorm + Linear + GELU through the panel kernel (E = 384); `variant` must be 0 (kept in the signature: ABI 4).
extern "C" int parseq_op_ln_linear_gelu(const float* x, const float* gamma, const float* beta, const void* W, const float* bias,
                                        void* out, int M, int N, int variant, void* stream) {
    CHK(check_arch());
    if (!x || !gamma || !beta || !W || !bias || !out || M <= 0 || N <= 0 || (N % PN_BN)) return fail(PARSEQ_E_INVALID, "bad argument (N must be a multiple of 128)");
    PanelGelu pg; pg.out = (bf16_t*)out; pg.ldo = N;
    hipStream_t s = (hipStream_t)stream;
    if (variant != 0) return fail(PARSEQ_E_INVALID, "variant %d (the ablation variants were removed)", variant);
    HIPCHK((launch_ln_panel_gemm<384, PanelGelu>(s, x, gamma, beta, 1e-6f, (const bf16_t*)W, bias, M, N, pg)));
    return 0;
}

extern "C" int parseq_op_encoder_attention(const void* q, const void* k, const void* vt, void* out, int dtype, int bh, int heads, void* stream) {
    CHK(check_arch());
    if (!q || !k || !vt || !out || bh <= 0 || heads <= 0 || bh % heads) return fail(PARSEQ_E_INVALID, "bad argument");
    if (dtype == PARSEQ_BF16) return run_enc_attention<bf16_t>((hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, bh, heads);
    SplitScope ss(dtype == PARSEQ_BF16X3);
    return run_enc_attention<float>((hipStream_t)stream, (const float*)q, (const float*)k, (const float*)vt, (float*)out, bh, heads);
}

