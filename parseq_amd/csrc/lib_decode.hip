// libparseq_hip.so — launch orchestration of the decoder passes, the AR loop, refinement and parseq_forward.
#include "lib_internal.h"

// -------------------------------------------------------------------------------------------------------------------
// decoder
// -------------------------------------------------------------------------------------------------------------------
// Cross-attention of Lq queries per image against the plan's cached memory K / V: tuned kernels for 128 memory tokens
// (streaming AR kernel, MFMA multi-query kernel), the key-count-generic kernel otherwise.
template <typename T, int E>
static int run_cross_attention(parseq_plan* p, hipStream_t s, int B, int Lq, float scale, T* ca, const QAsm qa = QAsm{}) {
    const int H = p->m->cfg.dec_heads, NK = p->m->tokens;
    const T* kmem = reinterpret_cast<const T*>(p->kmem); const T* vmem = reinterpret_cast<const T*>(p->vmem);
    const float* qc_ = p->qc;
    if (qa.nsplit && (qa.nsplit != DS_QS || NK != 128 || Lq != 1)) return fail(PARSEQ_E_STATE, "cross-attention: split q-projection outside the AR step kernels");
    if (NK != 128) {
        if constexpr (sizeof(T) == 2) {
            const int nt16 = (NK + 15) / 16;
#define PQ_CAM_N(NT)                                                                                                                      \
            if (nt16 > NT - 2 && nt16 <= NT) {                                                                                           \
                static LdsAttr attr_;                                                                                                    \
                HIPCHK(attr_.ensure(reinterpret_cast<const void*>(dec_cross_attn_mfma_n_kernel<NT>), dec_cross_attn_mfma_n_lds<NT>()));  \
                hipLaunchKernelGGL((dec_cross_attn_mfma_n_kernel<NT>), dim3((B * H + 1) / 2), dim3(128), dec_cross_attn_mfma_n_lds<NT>(), s, \
                                   qc_, kmem, vmem, H, Lq, NK, scale, ca, B * H);                                                      \
                HIPCHK(hipGetLastError());                                                                                                \
                return 0;                                                                                                                 \
            }
            PQ_CAM_N(2) PQ_CAM_N(4) PQ_CAM_N(6) PQ_CAM_N(8) PQ_CAM_N(10) PQ_CAM_N(12) PQ_CAM_N(14) PQ_CAM_N(16)
#undef PQ_CAM_N
        }
        const size_t lds = dec_cross_attn_generic_lds(NK);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(dec_cross_attn_generic_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((dec_cross_attn_generic_kernel<T>), dim3(B * H), dim3(128), lds, s, qc_, kmem, vmem, H, Lq, NK, scale, ca);
    } else if (Lq == 1) {
        if constexpr (sizeof(T) == 4 && E == 384) {
            if (p->kv24) {
                const auto kern = qa.nsplit ? dec_cross_attn_ar24_kernel<E, true> : dec_cross_attn_ar24_kernel<E, false>;
                hipLaunchKernelGGL(kern, dim3(B), dim3(E), 0, s, qc_, qa, reinterpret_cast<const unsigned char*>(p->kmem),
                                   reinterpret_cast<const unsigned char*>(p->vmem), p->kv_plane_elems, scale, ca);
                HIPCHK(hipGetLastError());
                return 0;
            }
        }
        if (p->kv24) return fail(PARSEQ_E_STATE, "cross-attention: 24-bit K / V rows but no kernel for this geometry");
        auto kern = dec_cross_attn_ar_kernel<T, E, false>;
        if constexpr (E <= 384) { if (qa.nsplit) kern = dec_cross_attn_ar_kernel<T, E, true>; }      // the split step exists for the fused AR loop's widths only
        else if (qa.nsplit) return fail(PARSEQ_E_STATE, "cross-attention: split q-projection at embed_dim %d", E);
        hipLaunchKernelGGL(kern, dim3(B), dim3(E), 0, s, qc_, qa, kmem, vmem, scale, ca);
    } else if constexpr (sizeof(T) == 2) {
        hipLaunchKernelGGL(dec_cross_attn_multi_mfma_kernel, dim3((B * H + 3) / 4), dim3(256), 0, s, qc_, kmem, vmem, H, Lq, scale, ca, B * H);
    } else {
        if (g_split && p->kv24)      // bf16x3: the matrix-core kernel on bf16 pairs, K / V from the 24-bit rows of the one-launch encoder's tail
            hipLaunchKernelGGL(dec_cross_attn_multi_mfma_x3_kernel<true>, dim3((B * H + 1) / 2), dim3(128), 0, s, qc_, kmem, vmem, H, Lq, scale, ca, B * H, p->kv_plane_elems);
        else if (g_split)
            hipLaunchKernelGGL(dec_cross_attn_multi_mfma_x3_kernel<false>, dim3((B * H + 1) / 2), dim3(128), 0, s, qc_, kmem, vmem, H, Lq, scale, ca, B * H, (size_t)0);
        else if (p->kv24) return fail(PARSEQ_E_STATE, "cross-attention: 24-bit K / V rows outside the bf16x3 mode");
        else
            hipLaunchKernelGGL((dec_cross_attn_multi_kernel<T>), dim3(B * H), dim3(128), 0, s, qc_, kmem, vmem, H, Lq, scale, ca);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

template <typename T, int E>
static int decode_pass_e(parseq_plan* p, hipStream_t s, int B, int Lk, int i0, int Lq, const unsigned char* qmask, const unsigned char* kpm,
                         float* logits, int Ltot, int argmax_mode, bool keep_t = false, const float* user_query = nullptr) {
    const parseq_model* m = p->m;
    const parseq_config& c = m->cfg;
    const int M = B * Lq, Fd = E * c.dec_mlp_ratio, C = m->classes, npos = c.max_label_length + 1, H = c.dec_heads;
    const Weights<T> W = weights_of<T>(p);
    const std::string d = "decoder.layers.0.";
    T* sa = reinterpret_cast<T*>(p->sa); T* ca = reinterpret_cast<T*>(p->ca); T* hdn = reinterpret_cast<T*>(p->hdn);
    const T* kmem = reinterpret_cast<const T*>(p->kmem); const T* vmem = reinterpret_cast<const T*>(p->vmem);
    const float scale = sqrtf(1.0f / (float)DEC_HD);
    int* eos_rows = p->counters; int* ar_len = p->counters + 1;
    if constexpr (sizeof(T) == 2 && E <= 384) {
        // AR step (one unmasked query per image): two fused row-block kernels around the cross-attention (decoder_step.h)
        if (Lq == 1 && !qmask && !kpm && C <= 128 && p->fused_step && p->wstep[0] && !keep_t && !user_query) {
            const dim3 grid((M + DS_ROWS - 1) / DS_ROWS), block(64 * DS_NW);
            static LdsAttr attr_pre, attr_post;
            HIPCHK(attr_pre.ensure(reinterpret_cast<const void*>(dec_step_pre_kernel<E>), dec_step_pre_lds<E>()));
            HIPCHK(attr_post.ensure(reinterpret_cast<const void*>(dec_step_post_kernel<E>), dec_step_post_lds<E>()));
            {
                ProfScope ps_(&p->prof, T_DEC_PRE, s);
                hipLaunchKernelGGL((dec_step_pre_kernel<E>), grid, block, dec_step_pre_lds<E>(), s, p->stab, reinterpret_cast<const bf16_t*>(p->kvtab),
                                   p->tok, LDT, c.num_tokens, npos, Lk, i0, p->wstep[0], m->p(d + "self_attn.out_proj.bias"),
                                   m->p("pos_queries") + (size_t)i0 * E, m->p(d + "norm1.weight"), m->p(d + "norm1.bias"), c.dec_ln_eps,
                                   p->wstep[1], m->p(d + "cross_attn.in_proj_bias"), p->t, p->qc, M);
                HIPCHK(hipGetLastError());
            }
            {
                ProfScope ps_(&p->prof, T_DEC_CA, s);
                CHK((run_cross_attention<T, E>(p, s, B, 1, scale, ca)));
            }
            {
                ProfScope ps_(&p->prof, T_DEC_POST, s);
                hipLaunchKernelGGL((dec_step_post_kernel<E>), grid, block, dec_step_post_lds<E>(), s, reinterpret_cast<const bf16_t*>(ca), p->t,
                                   p->wstep[2], m->p(d + "cross_attn.out_proj.bias"), m->p(d + "norm2.weight"),
                                   m->p(d + "norm2.bias"), p->wstep[3], m->p(d + "linear1.bias"), p->wstep[4],
                                   m->p(d + "linear2.bias"), m->p("decoder.norm.weight"), m->p("decoder.norm.bias"), c.dec_ln_eps,
                                   p->wstep[5], m->p("head.bias"), C, logits, Ltot, i0, M, argmax_mode, p->tok, LDT, c.eos_id,
                                   p->eos_seen, eos_rows, ar_len);
                HIPCHK(hipGetLastError());
            }
            return 0;
        }
    }
    // self-attention from the tables, out-projection, residual onto the raw position queries
    if (user_query) {
        // model.py:100-102 with a caller-supplied tgt_query [B, Lq, E]: q-projection of norm_q(query) at run time (the position-query
        // tables do not apply), scores against the content-key table, residual onto the caller's query itself
        const float qscale = sqrtf(1.0f / (float)DEC_HD);
        { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_ln_gemm<T, E>(s, user_query, m->p(d + "norm_q.weight"), m->p(d + "norm_q.bias"), c.dec_ln_eps,
                         W.w(d + "self_attn.in_proj_weight"), M, E, epi_store<float>(M, E, m->p(d + "self_attn.in_proj_bias"), p->qc, E, qscale), p->tn))); }
        {
            ProfScope ps_(&p->prof, T_DEC_SA, s);
            hipLaunchKernelGGL((dec_self_attn_kernel<T, E>), dim3(M), dim3(E), 0, s, p->stab, reinterpret_cast<const T*>(p->kvtab), p->tok, LDT,
                               c.num_tokens, npos, qmask, LDT, kpm, LDT, Lk, i0, Lq, sa, p->qc);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(p->t, user_query, (size_t)M * E * sizeof(float), hipMemcpyDeviceToDevice, s));
        { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_gemm<T>(s, ARowMajor<T>{sa, E}, W.w(d + "self_attn.out_proj.weight"), E, M, E, E,
                         epi_resid(M, E, m->p(d + "self_attn.out_proj.bias"), p->t, E)))); }
    } else {
    {
        ProfScope ps_(&p->prof, T_DEC_SA, s);
        if constexpr (sizeof(T) == 2 && E <= 512)
            hipLaunchKernelGGL((dec_self_attn_wave_kernel<E>), dim3((M + 3) / 4), dim3(256), 0, s, p->stab, reinterpret_cast<const bf16_t*>(p->kvtab),
                               p->tok, LDT, c.num_tokens, npos, qmask, LDT, kpm, LDT, Lk, i0, Lq, reinterpret_cast<bf16_t*>(sa), M);
        else {
            bool wave_form = false;
            if constexpr (sizeof(T) == 4 && E <= 512) wave_form = g_split;      // bf16x3: the same wave-per-row form on the f32 tables
            if constexpr (sizeof(T) == 4 && E <= 512) {
                if (wave_form)
                    hipLaunchKernelGGL((dec_self_attn_wave_kernel<E, float>), dim3((M + 3) / 4), dim3(256), 0, s, p->stab, reinterpret_cast<const float*>(p->kvtab),
                                       p->tok, LDT, c.num_tokens, npos, qmask, LDT, kpm, LDT, Lk, i0, Lq, reinterpret_cast<float*>(sa), M);
            }
            if (!wave_form)
                hipLaunchKernelGGL((dec_self_attn_kernel<T, E>), dim3(M), dim3(E), 0, s, p->stab, reinterpret_cast<const T*>(p->kvtab), p->tok, LDT,
                                   c.num_tokens, npos, qmask, LDT, kpm, LDT, Lk, i0, Lq, sa);
        }
        HIPCHK(hipGetLastError());
    }
    { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_gemm<T>(s, ARowMajor<T>{sa, E}, W.w(d + "self_attn.out_proj.weight"), E, M, E, E,
                     epi_table(M, E, m->p(d + "self_attn.out_proj.bias"), p->t, E, m->p("pos_queries"), E, Lq, i0)))); }
    }
    // cross-attention against memory (head-split K / V^T cached in the plan); norm1 is fused into the q-projection's A operand
    { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_ln_gemm<T, E>(s, p->t, m->p(d + "norm1.weight"), m->p(d + "norm1.bias"), c.dec_ln_eps,
                     W.w(d + "cross_attn.in_proj_weight"), M, E, epi_store<float>(M, E, m->p(d + "cross_attn.in_proj_bias"), p->qc, E), p->tn))); }
    {
        ProfScope ps_(&p->prof, T_DEC_CA, s);
        CHK((run_cross_attention<T, E>(p, s, B, Lq, scale, ca)));
    }
    { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_gemm<T>(s, ARowMajor<T>{ca, E}, W.w(d + "cross_attn.out_proj.weight"), E, M, E, E, epi_resid(M, E, m->p(d + "cross_attn.out_proj.bias"), p->t, E)))); }
    // MLP (norm2 fused into linear1's A operand)
    { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_ln_gemm<T, E>(s, p->t, m->p(d + "norm2.weight"), m->p(d + "norm2.bias"), c.dec_ln_eps,
                     W.w(d + "linear1.weight"), M, Fd, epi_gelu<T>(M, Fd, m->p(d + "linear1.bias"), hdn, Fd), p->tn))); }
    { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_gemm<T>(s, ARowMajor<T>{hdn, Fd}, W.w(d + "linear2.weight"), Fd, M, E, Fd, epi_resid(M, E, m->p(d + "linear2.bias"), p->t, E)))); }
    // decoder.norm fused into the head's A operand
    { ProfScope ps_(&p->prof, T_DEC_GEMM, s); CHK((run_ln_gemm<T, E>(s, p->t, m->p("decoder.norm.weight"), m->p("decoder.norm.bias"), c.dec_ln_eps,
                     W.w("head.weight"), M, C, epi_store<float>(M, C, m->p("head.bias"), logits, C, 1.f, Lq, Ltot, i0), p->tn))); }
    if (argmax_mode) {       // only meaningful for Lq == 1: greedy pick of position i0 into tok[:, i0 + 1]
        hipLaunchKernelGGL(ar_argmax_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, Ltot, C, p->tok, LDT, i0, B, c.eos_id,
                           p->eos_seen, eos_rows, ar_len, argmax_mode == 2 ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// One pass of the query stream (modules.py:55-98 with update_content=False, then decoder.norm and head) for queries
// pos_queries[i0 : i0 + Lq] of every image against the content tokens p->tok[:, :Lk].  Writes
// logits[b][i0 + qi][:] for qi < Lq into a [B][Ltot][C] tensor.
template <typename T>
static int decode_pass(parseq_plan* p, hipStream_t s, int B, int Lk, int i0, int Lq, const unsigned char* qmask, const unsigned char* kpm,
                       float* logits, int Ltot, int argmax_mode = 0, bool keep_t = false, const float* user_query = nullptr) {
    switch (p->m->cfg.embed_dim) {
        case 192: return decode_pass_e<T, 192>(p, s, B, Lk, i0, Lq, qmask, kpm, logits, Ltot, argmax_mode, keep_t, user_query);
        case 384: return decode_pass_e<T, 384>(p, s, B, Lk, i0, Lq, qmask, kpm, logits, Ltot, argmax_mode, keep_t, user_query);
        default:  return decode_pass_e<T, 768>(p, s, B, Lk, i0, Lq, qmask, kpm, logits, Ltot, argmax_mode, keep_t, user_query);
    }
}

// The whole AR loop with the mid / cross-attention / mlp arrangement of decoder_step.h (bf16 or, X3, the bf16x3 arithmetic on f32
// storage; E <= 384): step i's logits are produced by the mid kernel of step i + 1 (and by one trailing finish-only launch after
// the last step).
template <int E, bool X3 = false>
static int ar_loop_fused(parseq_plan* p, hipStream_t s, int B, int num_steps, float* logits, bool testing, bool latency) {
    const parseq_model* m = p->m;
    const parseq_config& c = m->cfg;
    const int M = B, C = m->classes, npos = c.max_label_length + 1;
    const std::string d = "decoder.layers.0.";
    int* eos_rows = p->counters; int* ar_len = p->counters + 1;
    using TS = typename std::conditional<X3, float, bf16_t>::type;      // storage type of kvtab, the memory K / V and ca
    TS* ca = reinterpret_cast<TS*>(p->ca);
    float* partial = reinterpret_cast<float*>(p->hdn);                    // linear2 partial sums [ds_split][M][E] f32 (the generic path's MLP hidden buffer is idle here)
    float* tq = p->qc;                                                    // t' lives in the q-projection buffer once the cross-attention has consumed it
    float* t = p->t;
    int* tok = p->tok;
    unsigned char* eos_seen = p->eos_seen;
    const float scale = sqrtf(1.0f / (float)DEC_HD);
    const dim3 grid((M + DS_ROWS - 1) / DS_ROWS), block(64 * DS_NW);
    static LdsAttr attr_mid, attr_midq, attr_mlp;
    HIPCHK(attr_mid.ensure(reinterpret_cast<const void*>(dec_step_mid_kernel<E, X3>), dec_step_mid_lds<E, X3>()));
    HIPCHK(attr_midq.ensure(reinterpret_cast<const void*>(dec_step_mid_kernel<E, X3, DS_QS>), dec_step_mid_lds<E, X3>()));
    HIPCHK(attr_mlp.ensure(reinterpret_cast<const void*>(dec_step_mlp_kernel<E, X3>), dec_step_mlp_lds<E, X3>()));
    // The start half of the mid kernel split over DS_QS workgroups per row tile (decoder_step.h): the q-projection arrives at the
    // cross-attention as partial sums behind t' in the q buffer ([M][E] t' | [DS_QS][M][E] partials | [DS_QS][M][2] column sums; the buffer
    // holds npos rows per image).  Only the two AR cross-attention kernels know how to read that (128 memory tokens).
    // Taken when the caller says this forward is alone on the device (PARSEQ_FLAG_LATENCY): the wider step is a shorter chain but costs
    // about twice the compute-unit time, which batches in flight on other streams would rather have (profiles/r04_ar_step_timers.md).
    const bool qsplit = latency && p->qsplit && m->tokens == 128 && npos >= DS_QS + 2;
    float* qp = tq + (size_t)M * E;
    float* qstats = qp + (size_t)DS_QS * M * E;
    QAsm qa;
    if (qsplit) { qa.qp = qp; qa.stats = qstats; qa.cq = p->qfold + E; qa.bq2 = p->qfold + 2 * E; qa.nsplit = DS_QS; qa.M = M; qa.inv_e = 1.0f / (float)E; qa.eps = c.dec_ln_eps; }
    for (int i = 0; i <= num_steps; ++i) {
        const int do_finish = i > 0, do_start = i < num_steps;
        // the pick of position i - 1 feeds step i: needed while there is a step to start
        const int argmax_mode = do_start ? (testing ? 2 : 1) : 0;
        {
            ProfScope ps_(&p->prof, T_DEC_PRE, s);
            const auto mid = qsplit ? dec_step_mid_kernel<E, X3, DS_QS> : dec_step_mid_kernel<E, X3, 1>;
            hipLaunchKernelGGL(mid, dim3(grid.x * (qsplit ? DS_QS : 1)), block, (dec_step_mid_lds<E, X3>()), s, do_finish, do_start, i, M,
                               tq, partial, m->p(d + "linear2.bias"), m->p("decoder.norm.weight"), m->p("decoder.norm.bias"), c.dec_ln_eps,
                               p->wstep[5], m->p("head.bias"), C, logits, num_steps, argmax_mode, c.eos_id, eos_seen, eos_rows, ar_len,
                               p->stab, reinterpret_cast<const TS*>(p->kvtab), tok, LDT, c.num_tokens, npos, p->wstep[0],
                               m->p(d + "self_attn.out_proj.bias"), m->p("pos_queries"), m->p(d + "norm1.weight"), m->p(d + "norm1.bias"),
                               p->wstep[1], m->p(d + "cross_attn.in_proj_bias"), t, qsplit ? qp : tq, p->qfold, qstats);
            HIPCHK(hipGetLastError());
        }
        if (!do_start) break;
        {
            ProfScope ps_(&p->prof, T_DEC_CA, s);
            CHK((run_cross_attention<TS, E>(p, s, B, 1, scale, ca, qa)));
        }
        {
            ProfScope ps_(&p->prof, T_DEC_POST, s);
            hipLaunchKernelGGL((dec_step_mlp_kernel<E, X3>), dim3(grid.x * ds_split<E>()), block, (dec_step_mlp_lds<E, X3>()), s, ca, t, p->wstep[2],
                               m->p(d + "cross_attn.out_proj.bias"), m->p(d + "norm2.weight"), m->p(d + "norm2.bias"), c.dec_ln_eps,
                               p->wstep[3], m->p(d + "linear1.bias"), p->wstep[4], tq, partial, M);
            HIPCHK(hipGetLastError());
        }
    }
    return 0;
}

template <typename T>
static int forward_impl(parseq_plan* p, int B, int flags, int refine_iters, int num_steps, float* logits, int* out_len, hipStream_t s) {
    const parseq_model* m = p->m;
    const parseq_config& c = m->cfg;
    const int C = m->classes;
    const bool ar = flags & PARSEQ_FLAG_DECODE_AR, testing = flags & PARSEQ_FLAG_TESTING, latency = flags & PARSEQ_FLAG_LATENCY;
    int* eos_rows = p->counters; int* ar_len = p->counters + 1;
    hipLaunchKernelGGL(ar_init_kernel, dim3((B * LDT + 255) / 256), dim3(256), 0, s, p->tok, LDT, B, c.bos_id, c.pad_id, p->eos_seen, p->counters, 2, num_steps);
    HIPCHK(hipGetLastError());
    if (ar) {
        // model.py:119-147.  All num_steps steps are always run (no per-step host sync); the step at which the reference
        // would have stopped is recorded on the device and only truncates the returned view.
        bool done = false;
        if constexpr (sizeof(T) == 2) {
            if (p->wstep[0] && p->fused_step && C <= 128 && c.dec_mlp_ratio == 4) {
                if (c.embed_dim == 384) { CHK((ar_loop_fused<384>(p, s, B, num_steps, logits, testing, latency))); done = true; }
                else if (c.embed_dim == 192) { CHK((ar_loop_fused<192>(p, s, B, num_steps, logits, testing, latency))); done = true; }
            }
        } else {
            // bf16x3: the same fused step on bf16 pairs (f32 tables, f32 memory K / V); the fp32 mode keeps the per-op kernels
            if (p->precision == PARSEQ_BF16X3 && p->wstep[0] && p->fused_step && C <= 128 && c.dec_mlp_ratio == 4) {
                if (c.embed_dim == 384) { CHK((ar_loop_fused<384, true>(p, s, B, num_steps, logits, testing, latency))); done = true; }
                else if (c.embed_dim == 192) { CHK((ar_loop_fused<192, true>(p, s, B, num_steps, logits, testing, latency))); done = true; }
            }
        }
        for (int i = 0; !done && i < num_steps; ++i) {
            // greedy pick of position i into tok[:, i + 1] (+ EOS bookkeeping) rides on the step; the last step needs none
            CHK((decode_pass<T>(p, s, B, i + 1, i, 1, nullptr, nullptr, logits, num_steps, i + 1 < num_steps ? (testing ? 2 : 1) : 0)));
        }
    } else {
        // model.py:148-152: context is <bos> only, all positions queried at once
        CHK((decode_pass<T>(p, s, B, 1, 0, num_steps, nullptr, nullptr, logits, num_steps)));
    }
    for (int it = 0; it < refine_iters; ++it) {
        // model.py:154-167
        // the first refinement after an AR decode: tok[:, 1:] already holds the greedy picks of positions 0 .. L-2 (the loop computed
        // them from these very logits), so the 95-wide arg-max scan per position (30 us of strided reads at batch 512) is skipped
        const int from_logits = (ar && it == 0) ? 0 : 1;
        hipLaunchKernelGGL(refine_prep_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, num_steps, C, p->tok, LDT, p->kpm, LDT, B, c.bos_id, c.eos_id, from_logits);
        HIPCHK(hipGetLastError());
        CHK((decode_pass<T>(p, s, B, num_steps, 0, num_steps, p->cloze, p->kpm, logits, num_steps)));
    }
    int L = num_steps;
    if (ar && testing && refine_iters == 0) {
        // the reference stops after the first step at which EVERY row holds an EOS: the device recorded that step (num_steps if it
        // never happened)
        int cnt[2];
        HIPCHK(hipMemcpyAsync(cnt, p->counters, sizeof(cnt), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        L = cnt[1];
    }
    if (out_len) *out_len = L;
    return 0;
}

// ViTSTR: strhub/models/vitstr/system.py:76-82 + vitstr/model.py:20-28.
extern "C" int parseq_vitstr_forward(parseq_plan* p, const void* images, int images_dtype, int batch, int num_steps, float* logits_out,
                                     void* stream) {
    CHK(check_call(p, batch, images_dtype));
    if (!p->m->vitstr) return fail(PARSEQ_E_INVALID, "parseq_vitstr_forward on a PARSeq model (arch 0)");
    DevGuard dg(p->m->device);
    SplitScope ss(p->precision == PARSEQ_BF16X3);
    if (!images || !logits_out) return fail(PARSEQ_E_INVALID, "null images / logits_out");
    const parseq_model* m = p->m;
    const int npos = m->cfg.max_label_length + 1, N = m->tokens, C = m->classes, E = m->cfg.embed_dim;
    if (num_steps < 1 || num_steps > npos) return fail(PARSEQ_E_INVALID, "num_steps %d outside [1, %d]", num_steps, npos);
    hipStream_t s = (hipStream_t)stream;
    CHK(encode_dispatch(p, images, images_dtype, batch, nullptr, s));
    // model.forward(images, seqlen = num_steps + 1): head over the first seqlen tokens, then [:, 1:] drops the class-token position.
    // The head runs over every token row of the batch (one plain GEMM on the normalised features); the wanted rows are sliced out.
    float* all = reinterpret_cast<float*>(p->h);           // [batch * N][C] scratch (the MLP hidden buffer is idle here)
    const int M = batch * N;
    if (p->precision == PARSEQ_BF16) {
        const Weights<bf16_t> W = weights_of<bf16_t>(p);
        CHK((run_gemm<bf16_t>(s, ARowMajor<bf16_t>{reinterpret_cast<const bf16_t*>(p->xn), E}, W.w("head.weight"), E, M, C, E, epi_store<float>(M, C, m->p("head.bias"), all, C))));
    } else {
        const Weights<float> W = weights_of<float>(p);
        CHK((run_gemm<float>(s, ARowMajor<float>{reinterpret_cast<const float*>(p->xn), E}, W.w("head.weight"), E, M, C, E, epi_store<float>(M, C, m->p("head.bias"), all, C))));
    }
    HIPCHK(hipMemcpy2DAsync(logits_out, (size_t)num_steps * C * sizeof(float), all + (size_t)C, (size_t)N * C * sizeof(float),
                            (size_t)num_steps * C * sizeof(float), batch, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int parseq_forward(parseq_plan* p, const void* images, int images_dtype, int batch, int flags, int refine_iters,
                              int num_steps, float* logits_out, int* out_len, void* stream) {
    CHK(check_call(p, batch, images_dtype));
    if (p->m->vitstr) return fail(PARSEQ_E_INVALID, "parseq_forward on a ViTSTR model: use parseq_vitstr_forward");
    DevGuard dg(p->m->device);
    SplitScope ss(p->precision == PARSEQ_BF16X3);
    if (!images || !logits_out) return fail(PARSEQ_E_INVALID, "null images / logits_out");
    const int npos = p->m->cfg.max_label_length + 1;
    if (num_steps < 1 || num_steps > npos) return fail(PARSEQ_E_INVALID, "num_steps %d outside [1, %d]", num_steps, npos);
    if (refine_iters < 0) return fail(PARSEQ_E_INVALID, "refine_iters %d", refine_iters);
    hipStream_t s = (hipStream_t)stream;
    CHK(encode_dispatch(p, images, images_dtype, batch, nullptr, s));
    // (A decoder stream of its own with hipStreamCreateWithPriority(greatest), forked and joined by events, was measured and removed:
    // 121 -> 108 k img/s with two forwards in flight, 107 -> 58 k one at a time — profiles/r03_decoder_priority_stream_ab.md.)
    if (p->precision == PARSEQ_BF16) return forward_impl<bf16_t>(p, batch, flags, refine_iters, num_steps, logits_out, out_len, s);
    return forward_impl<float>(p, batch, flags, refine_iters, num_steps, logits_out, out_len, s);
}

static int decode_entry(parseq_plan* p, const int32_t* tokens, int batch, int ctx_len, int q_start, int q_len, const uint8_t* query_mask,
                        const uint8_t* key_padding_mask, float* logits_out, float* hidden_out, void* stream, const float* user_query = nullptr) {
    if (!p || !tokens || !logits_out) return fail(PARSEQ_E_INVALID, "null argument");
    if (p->m->vitstr) return fail(PARSEQ_E_INVALID, "ViTSTR has no decoder");
    DevGuard dg(p->m->device);
    SplitScope ss(p->precision == PARSEQ_BF16X3);
    if (batch <= 0 || batch > p->max_batch || batch != p->last_batch) return fail(PARSEQ_E_INVALID, "batch %d does not match the last parseq_encode (%d)", batch, p->last_batch);
    const int npos = p->m->cfg.max_label_length + 1;
    if (ctx_len < 1 || ctx_len > npos || q_start < 0 || q_len < 1 || q_start + q_len > npos) return fail(PARSEQ_E_INVALID, "bad context / query range");
    hipStream_t s = (hipStream_t)stream;
    // stage caller's tokens / masks into the plan's pitched arrays
    HIPCHK(hipMemcpy2DAsync(p->tok, LDT * sizeof(int), tokens, ctx_len * sizeof(int), ctx_len * sizeof(int), batch, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(clamp_tokens_kernel, dim3((batch * ctx_len + 255) / 256), dim3(256), 0, s, p->tok, LDT, batch, ctx_len, p->m->cfg.num_tokens);
    HIPCHK(hipGetLastError());
    const unsigned char* kpm = nullptr; const unsigned char* qm = nullptr;
    if (key_padding_mask) {
        HIPCHK(hipMemcpy2DAsync(p->kpm, LDT, key_padding_mask, ctx_len, ctx_len, batch, hipMemcpyDeviceToDevice, s));
        kpm = p->kpm;
    }
    if (query_mask) {      // rows are relative to q_start; the kernel indexes by absolute query position
        HIPCHK(hipMemcpy2DAsync(p->qmask_user + (size_t)q_start * LDT, LDT, query_mask, ctx_len, ctx_len, q_len, hipMemcpyDeviceToDevice, s));
        qm = p->qmask_user;
    }
    const bool keep_t = hidden_out != nullptr;
    // decode_pass writes logits[b][q_start + qi] of a [B][Ltot][C] tensor (the forward's layout).  Here the caller's tensor is
    // [batch][q_len][C] with row qi: hand over the base shifted back by q_start rows, so that the rows written are exactly
    // [b * q_len + qi] (writing at b * q_len + q_start + qi ran q_start rows past the end of the buffer for q_start > 0).
    float* lbase = logits_out - (size_t)q_start * p->m->classes;
    if (p->precision == PARSEQ_BF16) CHK((decode_pass<bf16_t>(p, s, batch, ctx_len, q_start, q_len, qm, kpm, lbase, q_len, 0, keep_t, user_query)));
    else CHK((decode_pass<float>(p, s, batch, ctx_len, q_start, q_len, qm, kpm, lbase, q_len, 0, keep_t, user_query)));
    if (hidden_out) {      // model.decode's return value: decoder.norm of the query stream (modules.py:124), fp32
        const parseq_model* m = p->m;
        CHK((run_layernorm<float>(s, p->t, m->p("decoder.norm.weight"), m->p("decoder.norm.bias"), hidden_out, nullptr, batch * q_len,
                                  m->cfg.embed_dim, m->cfg.dec_ln_eps)));
    }
    return 0;
}

extern "C" int parseq_decode_logits(parseq_plan* p, const int32_t* tokens, int batch, int ctx_len, int q_start, int q_len,
                                    const uint8_t* query_mask, const uint8_t* key_padding_mask, float* logits_out, void* stream) {
    return decode_entry(p, tokens, batch, ctx_len, q_start, q_len, query_mask, key_padding_mask, logits_out, nullptr, stream);
}

extern "C" int parseq_decode_hidden(parseq_plan* p, const int32_t* tokens, int batch, int ctx_len, int q_start, int q_len,
                                    const uint8_t* query_mask, const uint8_t* key_padding_mask, float* hidden_out, float* logits_out,
                                    void* stream) {
    if (!hidden_out) return fail(PARSEQ_E_INVALID, "null hidden_out");
    return decode_entry(p, tokens, batch, ctx_len, q_start, q_len, query_mask, key_padding_mask, logits_out, hidden_out, stream);
}

extern "C" int parseq_decode_query(parseq_plan* p, const int32_t* tokens, int batch, int ctx_len, const float* query, int q_len,
                                   const uint8_t* query_mask, const uint8_t* key_padding_mask, float* hidden_out, float* logits_out,
                                   void* stream) {
    if (!query) return fail(PARSEQ_E_INVALID, "null query");
    if (!p) return fail(PARSEQ_E_INVALID, "null plan");
    if (q_len < 1 || q_len > p->m->cfg.max_label_length + 1) return fail(PARSEQ_E_INVALID, "q_len %d outside [1, %d]", q_len, p->m->cfg.max_label_length + 1);
    return decode_entry(p, tokens, batch, ctx_len, 0, q_len, query_mask, key_padding_mask, logits_out, hidden_out, stream, query);
}

// The cross-attention K / V of a caller-supplied encoder output (model.decode's `memory` argument, model.py:89): replaces the
// K / V cached by the last parseq_encode on this plan.
template <typename T>
static int set_memory_impl(parseq_plan* p, const float* memory, int B, hipStream_t s) {
    const parseq_model* m = p->m;
    const parseq_config& c = m->cfg;
    const int E = c.embed_dim, N = m->tokens, M = B * N;
    const Weights<T> W = weights_of<T>(p);
    const T* a;
    if constexpr (sizeof(T) == 2) {
        const size_t n = (size_t)M * E;
        hipLaunchKernelGGL(cvt_f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, memory, reinterpret_cast<bf16_t*>(p->xn), n);
        HIPCHK(hipGetLastError());
        a = reinterpret_cast<const T*>(p->xn);
    } else {
        a = memory;
    }
    const std::string d = "decoder.layers.0.cross_attn.";
    EpiHeads<T> ek; static_cast<EpiBase&>(ek) = epi_base(M, 2 * E, m->p(d + "in_proj_bias") + E);
    ek.seg[0] = reinterpret_cast<T*>(p->kmem); ek.seg[1] = reinterpret_cast<T*>(p->vmem); ek.seg[2] = nullptr;
    ek.E = E; ek.heads = c.dec_heads; ek.hd = DEC_HD; ek.tokens = N; ek.tr_from = 2;
    p->kv24 = false;      // rows in the storage type from the generic GEMM
    ProfScope ps_(&p->prof, T_KVMEM, s);
    CHK((run_gemm<T>(s, ARowMajor<T>{a, E}, W.w(d + "in_proj_weight") + (size_t)E * E, E, M, 2 * E, E, ek, E % 128 != 0)));
    p->last_batch = B;
    return 0;
}

extern "C" int parseq_set_memory(parseq_plan* p, const float* memory, int batch, void* stream) {
    if (!p || !memory) return fail(PARSEQ_E_INVALID, "null argument");
    if (p->m->vitstr) return fail(PARSEQ_E_INVALID, "ViTSTR has no decoder");
    if (batch <= 0 || batch > p->max_batch) return fail(PARSEQ_E_INVALID, "batch %d outside (0, %d]", batch, p->max_batch);
    if (p->packed_version != p->m->version) return fail(PARSEQ_E_STATE, "model parameters changed after the plan was packed; call parseq_plan_refresh");
    DevGuard dg(p->m->device);
    SplitScope ss(p->precision == PARSEQ_BF16X3);
    if (p->precision == PARSEQ_BF16) return set_memory_impl<bf16_t>(p, memory, batch, (hipStream_t)stream);
    return set_memory_impl<float>(p, memory, batch, (hipStream_t)stream);
}
