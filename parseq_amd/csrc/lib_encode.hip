// libparseq_hip.so — launch orchestration of the encoder: patch embedding, blocks, final norm, memory K / V (parseq_encode).
#include "lib_internal.h"

// -------------------------------------------------------------------------------------------------------------------
// encoder
// -------------------------------------------------------------------------------------------------------------------

template <typename T, typename TI>
static int encode_impl(parseq_plan* p, const TI* images, int B, float* memory_out, hipStream_t s) {
    const parseq_model* m = p->m;
    const parseq_config& c = m->cfg;
    const int E = c.embed_dim, N = m->tokens, M = B * N, H = c.enc_heads, F = E * c.enc_mlp_ratio;
    const Weights<T> W = weights_of<T>(p);
    T* xn = reinterpret_cast<T*>(p->xn); T* q = reinterpret_cast<T*>(p->q); T* k = reinterpret_cast<T*>(p->k);
    T* vt = reinterpret_cast<T*>(p->vt); T* ao = reinterpret_cast<T*>(p->ao); T* h = reinterpret_cast<T*>(p->h);

    // patch embedding (im2col-free) + bias + pos_embed -> x        timm PatchEmbed; forward_features `x + pos_embed`
    const std::string& pe = m->enc;
    const int Np = m->patch_tokens, Mp = B * Np;
    APatch<T, TI> ap{images, 3, c.img_h, c.img_w, c.patch_h, c.patch_w, c.img_w / c.patch_w, Np};
    // bf16, PARSeq-S geometry: the patch embedding is the head of the one-launch encoder (encoder_blocks.h patch_head); same conditions
    // as `fused_blocks` below plus the (4, 8)-patch / 32 x 128-crop layout the head is written for
    const bool head_in_launch = sizeof(T) == 2 && !m->vitstr && p->fused_head && p->fused_blocks && p->fused_attn && p->mlp_resident &&
                                E == 384 && c.enc_mlp_ratio == 4 && N == ATT_N && c.patch_h == 4 && c.patch_w == 8 && c.img_h == 32 && c.img_w == 128 &&
                                p->wpe_off >= EB_HEAD_MIN_WPE;     // see EB_HEAD_MIN_WPE (always true with pos_embed ahead of the weight)
    // the same for the bf16x3 one-launch encoder (encoder_blocks_x3.h patch_head_x3; conditions of its launch below)
    const bool one_launch_x3 = p->fused_x3 && B > p->small_batch_max;      // small batches: per-operation launches (lib_internal.h small_batch_max)
    const bool head_x3 = sizeof(T) == 4 && g_split && one_launch_x3 && p->fused_blocks && p->fused_head && !m->vitstr && E == 384 && c.enc_mlp_ratio == 4 &&
                         N == ATT_N && M % 128 == 0 && c.patch_h == 4 && c.patch_w == 8 && c.img_h == 32 && c.img_w == 128 && p->wpe_off >= x3::X3_HEAD_MIN_WPE;
    if (head_in_launch || head_x3) {
        // nothing here: x is produced inside the launch
    } else if (!m->vitstr) {
        ProfScope ps_(&p->prof, T_PATCH, s);
        CHK((run_gemm<T>(s, ap, W.w(pe + "patch_embed.proj.weight"), m->patch_k, Mp, E, m->patch_k,
                         epi_table(Mp, E, m->p(pe + "patch_embed.proj.bias"), p->x, E, m->p(pe + "pos_embed"), E, Np, 0))));
    } else {
        // ViTSTR (timm class_token=True): x[b] = [cls_token; patches] + pos_embed[0 .. Np].  The patch rows (with pos_embed[1..])
        // go to a scratch tile first (the idle MLP hidden buffer), then one pass interleaves the class-token rows
        float* xp = reinterpret_cast<float*>(p->h);
        ProfScope ps_(&p->prof, T_PATCH, s);
        CHK((run_gemm<T>(s, ap, W.w(pe + "patch_embed.proj.weight"), m->patch_k, Mp, E, m->patch_k,
                         epi_table(Mp, E, m->p(pe + "patch_embed.proj.bias"), xp, E, m->p(pe + "pos_embed") + E, E, Np, 0))));
        const size_t total4 = (size_t)M * E / 4;
        hipLaunchKernelGGL(insert_cls_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, xp, m->p("cls_token"), m->p(pe + "pos_embed"),
                           p->x, B, Np, E);
        HIPCHK(hipGetLastError());
    }
    // bf16 mode: LayerNorm + projection fused in the register-resident-A panel kernel (encoder_panel.h) wherever the
    // output width is a multiple of its 128-column tile; otherwise (and in f32 mode) LayerNorm kernel + generic tile GEMM.
    constexpr bool kBf16 = sizeof(T) == 2;
    const bool panel_qkv = kBf16 && (E == 192 || E == 384) && (3 * E) % PN_BN == 0;
    const bool panel_fc1 = kBf16 && (E == 192 || E == 384) && F % PN_BN == 0;
    const bool fused_mlp = kBf16 && E == 384 && c.enc_mlp_ratio == 4;      // encoder_mlp.h: LayerNorm + fc1 + GELU + fc2 + residual in one kernel
    // bf16x3, big M: activations travel between the encoder's kernels already split into block-planar hi | lo bf16 pairs — the
    // LayerNorm, the attention kernel and the fc1 + GELU epilogue write that form, and the four GEMMs of a block run the
    // direct-to-LDS loop on both operands (gemm.h PAIRS) instead of converting their A tile in every column tile's workgroup.
#ifndef PQ_X3_PRESPLIT
#define PQ_X3_PRESPLIT 1
#endif
    const bool presplit = PQ_X3_PRESPLIT && !kBf16 && g_split && M >= 4096 && N == ATT_N && (E == 384 || E == 768);
#ifndef PQ_X3_LN_IN_GEMM
#define PQ_X3_LN_IN_GEMM 0      // measured: qkv 270 + 48 us (LayerNorm launch) vs 319 us fused, fc1 360 + 48 vs 429: the loader's arithmetic costs what the launch did
#endif
    const bool ln_in_gemm = PQ_X3_LN_IN_GEMM && !kBf16 && g_split && M >= 4096 && (E == 192 || E == 384 || E == 768);
    // A kernel that holds `per_cu` workgroups of 128 rows per CU finishes in whole rounds of per_cu * CUs row tiles.  When
    // the row count leaves a few tiles over (ViTSTR: 512 x 129 rows = 516 tiles on 256 CUs), those tiles would cost a whole
    // extra round; instead the leading whole rounds go to the fused kernel and the tail rows to the generic kernels.
    auto main_rows = [&](int per_cu) {
        const int tiles = (M + 127) / 128, slots = per_cu * p->num_cus, rem = tiles % slots;
        return (tiles > slots && rem > 0 && rem <= slots / 16) ? (tiles - rem) * 128 : M;
    };
    // encoder_attn_fused.h: LayerNorm + qkv + attention + proj + residual in one kernel, one image (128 tokens) per workgroup
    // (one workgroup per image whatever the batch: with a partial last round of workgroups the fused kernels just run it — routing
    // those images through other kernels would make an image's result depend on its position in the batch)
    const bool fused_attn = kBf16 && E == 384 && N == ATT_N && p->fused_attn;
    const int Ma = fused_attn ? M : 0;
    const int Mq = panel_qkv ? main_rows(2) : M, Mm = fused_mlp ? main_rows(1) : M;
    // encoder_blocks.h: all blocks in ONE launch, x resident in registers from the first LayerNorm to the last residual
    const bool fused_blocks = fused_attn && fused_mlp && p->fused_blocks && p->mlp_resident && M % 128 == 0;
    if (fused_blocks) {
        if constexpr (kBf16) {
            // parseq_forward (nobody asked for `memory` itself): the final LayerNorm and the decoder's K / V projection of memory ride
            // in the same launch (encoder_blocks.h kv_phase) and the encoder is done
            const bool tail = p->fused_tail && memory_out == nullptr && !m->vitstr && c.dec_heads * DEC_HD == E;
            EncTailParams et = p->enc_tail;
            if (tail) { et.kmem = reinterpret_cast<bf16_t*>(p->kmem); et.vmem = reinterpret_cast<bf16_t*>(p->vmem); }
            EncHeadParams eh{nullptr, 0, 0, nullptr};
            if (head_in_launch) {
                eh.images = images; eh.img_dtype = sizeof(TI) == 1 ? EB_IMG_U8 : (sizeof(TI) == 2 ? EB_IMG_BF16 : EB_IMG_F32);
                eh.wpe = p->wpe_off; eh.posb = p->posb;
            }
            {
                ProfScope ps_(&p->prof, T_BLOCKS, s);
                HIPCHK((launch_enc_blocks<384>(s, p->x, reinterpret_cast<const bf16_t*>(p->wpack), m->master_elems * sizeof(bf16_t), m->master,
                                               p->blocks_dev, c.enc_depth, c.enc_ln_eps, M, et, eh)));
            }
            if (tail) { p->last_batch = B; return 0; }
        }
    }
    bool blocks_done = fused_blocks;
    if constexpr (!kBf16) {
        // bf16x3, PARSeq-S geometry: the twelve blocks — and, when nobody asked for `memory` itself, the final LayerNorm and the decoder's
        // K / V projection of it — in one launch with x resident in registers (encoder_blocks_x3w.h: eight waves of 16 rows; encoder_blocks_x3.h: four of 32); the MLP hidden buffer (idle on this
        // path) is the launch's per-image scratch (the parked residual stream and the attention output, 384 KiB per image)
        if (g_split && one_launch_x3 && p->fused_blocks && !m->vitstr && E == 384 && c.enc_mlp_ratio == 4 && N == ATT_N && M % 128 == 0) {
            const bool tail = p->fused_tail && memory_out == nullptr && c.dec_heads * DEC_HD == E;
            x3::EncTailX3 et{p->enc_tail.norm_w, p->enc_tail.norm_b, p->enc_tail.wkv, p->enc_tail.bkv, nullptr, nullptr, p->enc_tail.heads};
            if (tail) { et.kmem = reinterpret_cast<float*>(p->kmem); et.vmem = reinterpret_cast<float*>(p->vmem); }
            // the tail's K / V rows as 24-bit floats (3 bytes per element: decoder_attn.h F24) — the cross-attention kernels of this
            // geometry read either format, whichever the last producer left (p->kv24)
            if (tail && p->kv24_enabled && N == 128) et.plane_elems = p->kv_plane_elems;
            p->kv24 = tail && et.plane_elems != 0;
            {
                // PARSEQ_X3_SPLIT=n (diagnostics): the blocks in n launches of depth / n, x through HBM between them (100 MB out + in at batch
                // 512) — shorter persistent workgroups, for the A/B of how a second batch's decoder interleaves with this launch
                static const int split = [] { const char* e = getenv("PARSEQ_X3_SPLIT"); const int v = e ? atoi(e) : 1; return v >= 1 ? v : 1; }();
                ProfScope ps_(&p->prof, T_BLOCKS, s);
                const int per = (c.enc_depth + split - 1) / split;
                for (int l0 = 0; l0 < c.enc_depth; l0 += per) {
                    const int d = std::min(per, c.enc_depth - l0);
                    const bool last = l0 + d >= c.enc_depth;
                    x3::EncHeadX3 eh{nullptr, 0, 0, nullptr};
                    if (head_x3 && l0 == 0) {
                        eh.images = images; eh.img_dtype = sizeof(TI) == 1 ? EB_IMG_U8 : (sizeof(TI) == 2 ? EB_IMG_BF16 : EB_IMG_F32);
                        eh.wpe = p->wpe_off; eh.posb = p->posb;
                    }
                    const x3::EncTailX3 et_l = last ? et : x3::EncTailX3{0, 0, 0, 0, nullptr, nullptr, 0};
                    if (p->x3_four_waves)
                        HIPCHK((x3::launch_enc_blocks_x3<384>(s, p->x, p->wpack, m->master_elems * sizeof(float), m->master, p->blocks_dev + l0, d,
                                                              c.enc_ln_eps, M, reinterpret_cast<float*>(p->h), et_l, eh)));
                    else
                        HIPCHK((x3w::launch_enc_blocks_x3w<384>(s, p->x, p->wpack, m->master_elems * sizeof(float), m->master, p->blocks_dev + l0, d,
                                                                c.enc_ln_eps, M, reinterpret_cast<float*>(p->h), et_l, eh)));
                }
            }
            if (tail) { p->last_batch = B; return 0; }
            blocks_done = true;
        }
    }
    for (int i = 0; i < (blocks_done ? 0 : c.enc_depth); ++i) {
        const std::string b = pe + "blocks." + std::to_string(i) + ".";
        if (fused_attn && Ma == M) {
            if constexpr (kBf16) {
                ProfScope ps_(&p->prof, T_ATTNF, s);
                HIPCHK((launch_fused_attn<384>(s, p->x, m->p(b + "norm1.weight"), m->p(b + "norm1.bias"), c.enc_ln_eps, W.w(b + "attn.qkv.weight"),
                                               m->p(b + "attn.qkv.bias"), W.w(b + "attn.proj.weight"), m->p(b + "attn.proj.bias"), M)));
            }
        } else {
        if (panel_qkv) {
            if constexpr (kBf16) {
                PanelHeads ph; ph.seg[0] = q; ph.seg[1] = k; ph.seg[2] = vt; ph.E = E; ph.heads = H; ph.hd = ATT_HD; ph.tokens = N;
                ProfScope ps_(&p->prof, T_QKV, s);
                if (E == 384) HIPCHK((launch_ln_panel_gemm<384>(s, p->x, m->p(b + "norm1.weight"), m->p(b + "norm1.bias"), c.enc_ln_eps, W.w(b + "attn.qkv.weight"), m->p(b + "attn.qkv.bias"), Mq, 3 * E, ph)));
                else HIPCHK((launch_ln_panel_gemm<192>(s, p->x, m->p(b + "norm1.weight"), m->p(b + "norm1.bias"), c.enc_ln_eps, W.w(b + "attn.qkv.weight"), m->p(b + "attn.qkv.bias"), Mq, 3 * E, ph)));
                if (Mq < M) {        // tail rows: LayerNorm kernel + generic GEMM, same head-split row-major outputs
                    const int Mt = M - Mq;
                    CHK((run_layernorm<T>(s, p->x + (size_t)Mq * E, m->p(b + "norm1.weight"), m->p(b + "norm1.bias"), xn + (size_t)Mq * E, nullptr, Mt, E, c.enc_ln_eps)));
                    EpiHeads<T> eq; static_cast<EpiBase&>(eq) = epi_base(Mt, 3 * E, m->p(b + "attn.qkv.bias"));
                    eq.seg[0] = q; eq.seg[1] = k; eq.seg[2] = vt; eq.E = E; eq.heads = H; eq.hd = ATT_HD; eq.tokens = N; eq.tr_from = 3; eq.m_off = Mq;
                    CHK((run_gemm<T>(s, ARowMajor<T>{xn + (size_t)Mq * E, E}, W.w(b + "attn.qkv.weight"), E, Mt, 3 * E, E, eq)));
                }
            }
        } else {
            EpiHeads<T> eq; static_cast<EpiBase&>(eq) = epi_base(M, 3 * E, m->p(b + "attn.qkv.bias"));
            eq.seg[0] = q; eq.seg[1] = k; eq.seg[2] = vt; eq.E = E; eq.heads = H; eq.hd = ATT_HD; eq.tokens = N;
            eq.tr_from = N == ATT_N ? 2 : 3;      // the 128-token kernels of this path read V^T, the generic one row-major V
            if (presplit) {
                if constexpr (!kBf16) {
                    { ProfScope ps_(&p->prof, T_LN, s); CHK((run_layernorm_split(s, p->x, m->p(b + "norm1.weight"), m->p(b + "norm1.bias"), reinterpret_cast<unsigned char*>(xn), M, E, c.enc_ln_eps))); }
                    ProfScope ps_(&p->prof, T_QKV, s);
                    HIPCHK((launch_gemm_pairs<128, 128, 2, 2>(s, reinterpret_cast<const bf16_t*>(xn), 2 * E, reinterpret_cast<const bf16_t*>(W.w(b + "attn.qkv.weight")), 2 * E, M, 3 * E, 2 * E, eq)));
                }
            } else if (ln_in_gemm) {       // bf16x3: row statistics in a 12 us pass, the LayerNorm itself in the GEMM's A-loader (run_ln_gemm)
                ProfScope ps_(&p->prof, T_QKV, s);
                CHK((run_ln_gemm_e<T>(s, E, p->x, m->p(b + "norm1.weight"), m->p(b + "norm1.bias"), c.enc_ln_eps, W.w(b + "attn.qkv.weight"), M, 3 * E, eq, xn)));
            } else {
                { ProfScope ps_(&p->prof, T_LN, s); CHK((run_layernorm<T>(s, p->x, m->p(b + "norm1.weight"), m->p(b + "norm1.bias"), xn, nullptr, M, E, c.enc_ln_eps))); }
                { ProfScope ps_(&p->prof, T_QKV, s); CHK((run_gemm<T>(s, ARowMajor<T>{xn, E}, W.w(b + "attn.qkv.weight"), E, M, 3 * E, E, eq))); }
            }
        }
        { ProfScope ps_(&p->prof, T_ATTN, s); CHK((run_enc_attention<T>(s, q, k, vt, ao, B * H, H, panel_qkv || N != ATT_N, N, presplit))); }
        if (presplit) {
            ProfScope ps_(&p->prof, T_PROJ, s);
            HIPCHK((launch_gemm_pairs<128, 128, 2, 2>(s, reinterpret_cast<const bf16_t*>(ao), 2 * E, reinterpret_cast<const bf16_t*>(W.w(b + "attn.proj.weight")), 2 * E, M, E, 2 * E,
                                                        epi_resid(M, E, m->p(b + "attn.proj.bias"), p->x, E))));
        } else
        { ProfScope ps_(&p->prof, T_PROJ, s); CHK((run_gemm<T>(s, ARowMajor<T>{ao, E}, W.w(b + "attn.proj.weight"), E, M, E, E, epi_resid(M, E, m->p(b + "attn.proj.bias"), p->x, E)))); }
        }
        if (fused_mlp) {
            if constexpr (kBf16) {
                ProfScope ps_(&p->prof, T_MLP, s);
                if (p->mlp_resident)
                    HIPCHK((launch_fused_mlp<384, true>(s, p->x, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), c.enc_ln_eps, W.w(b + "mlp.fc1.weight"),
                                                           m->p(b + "mlp.fc1.bias"), W.w(b + "mlp.fc2.weight"), m->p(b + "mlp.fc2.bias"), Mm)));
                else
                HIPCHK((launch_fused_mlp<384>(s, p->x, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), c.enc_ln_eps, W.w(b + "mlp.fc1.weight"),
                                              m->p(b + "mlp.fc1.bias"), W.w(b + "mlp.fc2.weight"), m->p(b + "mlp.fc2.bias"), Mm)));
                if (Mm < M) {        // tail rows through the per-op kernels (same rounding points)
                    const int Mt = M - Mm;
                    CHK((run_layernorm<T>(s, p->x + (size_t)Mm * E, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), xn + (size_t)Mm * E, nullptr, Mt, E, c.enc_ln_eps)));
                    CHK((run_gemm<T>(s, ARowMajor<T>{xn + (size_t)Mm * E, E}, W.w(b + "mlp.fc1.weight"), E, Mt, F, E, epi_gelu<T>(Mt, F, m->p(b + "mlp.fc1.bias"), h + (size_t)Mm * F, F))));
                    CHK((run_gemm<T>(s, ARowMajor<T>{h + (size_t)Mm * F, F}, W.w(b + "mlp.fc2.weight"), F, Mt, E, F, epi_resid(Mt, E, m->p(b + "mlp.fc2.bias"), p->x + (size_t)Mm * E, E))));
                }
            }
            continue;
        }
        if (panel_fc1) {
            if constexpr (kBf16) {
                PanelGelu pg; pg.out = h; pg.ldo = F;
                ProfScope ps_(&p->prof, T_FC1, s);
                if (E == 384) HIPCHK((launch_ln_panel_gemm<384>(s, p->x, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), c.enc_ln_eps, W.w(b + "mlp.fc1.weight"), m->p(b + "mlp.fc1.bias"), M, F, pg)));
                else HIPCHK((launch_ln_panel_gemm<192>(s, p->x, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), c.enc_ln_eps, W.w(b + "mlp.fc1.weight"), m->p(b + "mlp.fc1.bias"), M, F, pg)));
            }
        } else {
            if (presplit) {
                if constexpr (!kBf16) {
                    { ProfScope ps_(&p->prof, T_LN, s); CHK((run_layernorm_split(s, p->x, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), reinterpret_cast<unsigned char*>(xn), M, E, c.enc_ln_eps))); }
                    EpiGeluSplit eg; static_cast<EpiBase&>(eg) = epi_base(M, F, m->p(b + "mlp.fc1.bias")); eg.out = reinterpret_cast<unsigned char*>(h); eg.ldo = F;
                    { ProfScope ps_(&p->prof, T_FC1, s);
                      HIPCHK((launch_gemm_pairs<128, 128, 2, 2>(s, reinterpret_cast<const bf16_t*>(xn), 2 * E, reinterpret_cast<const bf16_t*>(W.w(b + "mlp.fc1.weight")), 2 * E, M, F, 2 * E, eg))); }
                    ProfScope ps_(&p->prof, T_FC2, s);
                    HIPCHK((launch_gemm_pairs<128, 128, 2, 2>(s, reinterpret_cast<const bf16_t*>(h), 2 * F, reinterpret_cast<const bf16_t*>(W.w(b + "mlp.fc2.weight")), 2 * F, M, E, 2 * F,
                                                                epi_resid(M, E, m->p(b + "mlp.fc2.bias"), p->x, E))));
                }
                continue;
            } else if (ln_in_gemm) {
                ProfScope ps_(&p->prof, T_FC1, s);
                CHK((run_ln_gemm_e<T>(s, E, p->x, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), c.enc_ln_eps, W.w(b + "mlp.fc1.weight"), M, F, epi_gelu<T>(M, F, m->p(b + "mlp.fc1.bias"), h, F), xn)));
            } else {
                { ProfScope ps_(&p->prof, T_LN, s); CHK((run_layernorm<T>(s, p->x, m->p(b + "norm2.weight"), m->p(b + "norm2.bias"), xn, nullptr, M, E, c.enc_ln_eps))); }
                { ProfScope ps_(&p->prof, T_FC1, s); CHK((run_gemm<T>(s, ARowMajor<T>{xn, E}, W.w(b + "mlp.fc1.weight"), E, M, F, E, epi_gelu<T>(M, F, m->p(b + "mlp.fc1.bias"), h, F)))); }
            }
        }
        { ProfScope ps_(&p->prof, T_FC2, s); CHK((run_gemm<T>(s, ARowMajor<T>{h, F}, W.w(b + "mlp.fc2.weight"), F, M, E, F, epi_resid(M, E, m->p(b + "mlp.fc2.bias"), p->x, E)))); }
    }
    // final norm -> memory (fp32 to the caller, T copy as GEMM operand), then the cross-attention K/V of memory, ONCE
    { ProfScope ps_(&p->prof, T_LN, s); CHK((run_layernorm<T>(s, p->x, m->p(pe + "norm.weight"), m->p(pe + "norm.bias"), xn, memory_out, M, E, c.enc_ln_eps))); }
    p->last_batch = B;
    if (m->vitstr) return 0;          // no decoder: the head reads xn (parseq_vitstr_forward)
    const std::string d = "decoder.layers.0.cross_attn.";
    p->kv24 = false;      // f32 / bf16 rows from the generic GEMM
    {
        EpiHeads<T> ek; static_cast<EpiBase&>(ek) = epi_base(M, 2 * E, m->p(d + "in_proj_bias") + E);
        ek.seg[0] = reinterpret_cast<T*>(p->kmem); ek.seg[1] = reinterpret_cast<T*>(p->vmem); ek.seg[2] = nullptr;
        ek.E = E; ek.heads = c.dec_heads; ek.hd = DEC_HD; ek.tokens = N; ek.tr_from = 2;      // K and V both [b][h][key][32]
        ProfScope ps_(&p->prof, T_KVMEM, s);
        // the K | V boundary (column E) must fall on a tile edge: 64-wide tiles when E is not a multiple of 128 (PARSeq-Ti)
        CHK((run_gemm<T>(s, ARowMajor<T>{xn, E}, W.w(d + "in_proj_weight") + (size_t)E * E, E, M, 2 * E, E, ek, E % 128 != 0)));
    }
    p->last_batch = B;
    return 0;
}

int check_call(parseq_plan* p, int batch, int images_dtype) {
    if (!p) return fail(PARSEQ_E_INVALID, "null plan");
    if (batch <= 0 || batch > p->max_batch) return fail(PARSEQ_E_INVALID, "batch %d outside (0, %d]", batch, p->max_batch);
    if (images_dtype != PARSEQ_F32 && images_dtype != PARSEQ_BF16 && images_dtype != PARSEQ_U8) return fail(PARSEQ_E_INVALID, "images_dtype %d", images_dtype);
    if (p->packed_version != p->m->version) return fail(PARSEQ_E_STATE, "model parameters changed after the plan was packed; call parseq_plan_refresh");
    return 0;
}

int encode_dispatch(parseq_plan* p, const void* images, int images_dtype, int batch, float* memory_out, hipStream_t s) {
    if (p->precision == PARSEQ_BF16) {
        if (images_dtype == PARSEQ_F32) return encode_impl<bf16_t, float>(p, (const float*)images, batch, memory_out, s);
        if (images_dtype == PARSEQ_U8) return encode_impl<bf16_t, uint8_t>(p, (const uint8_t*)images, batch, memory_out, s);
        return encode_impl<bf16_t, bf16_t>(p, (const bf16_t*)images, batch, memory_out, s);
    }
    if (images_dtype == PARSEQ_F32) return encode_impl<float, float>(p, (const float*)images, batch, memory_out, s);
    if (images_dtype == PARSEQ_U8) return encode_impl<float, uint8_t>(p, (const uint8_t*)images, batch, memory_out, s);
    return encode_impl<float, bf16_t>(p, (const bf16_t*)images, batch, memory_out, s);
}

extern "C" int parseq_encode(parseq_plan* p, const void* images, int images_dtype, int batch, float* memory_out, void* stream) {
    CHK(check_call(p, batch, images_dtype));
    if (!images) return fail(PARSEQ_E_INVALID, "null images");
    DevGuard dg(p->m->device);
    SplitScope ss(p->precision == PARSEQ_BF16X3);
    return encode_dispatch(p, images, images_dtype, batch, memory_out, (hipStream_t)stream);
}

