// libparseq_hip.so — model and plan objects of the C ABI (include/parseq_hip.h): parameters, weight packs, decoder tables.
#include "lib_internal.h"

extern "C" int parseq_abi_version(void) { return PARSEQ_ABI_VERSION; }
extern "C" int parseq_shard_bounds(int64_t n, int world, int rank, int64_t* begin, int64_t* end) {
    if (!begin || !end) return fail(PARSEQ_E_INVALID, "null argument");
    if (n < 0 || world < 1 || rank < 0 || rank >= world) return fail(PARSEQ_E_INVALID, "shard %d of %d over %lld items", rank, world, (long long)n);
    const int64_t base = n / world, extra = n % world;
    *begin = rank * base + (rank < extra ? rank : extra);
    *end = *begin + base + (rank < extra ? 1 : 0);
    return 0;
}
extern "C" const char* parseq_last_error(void) { return g_err; }

extern "C" int parseq_model_create(const parseq_config* c, parseq_model** out) {
    if (!c || !out) return fail(PARSEQ_E_INVALID, "null argument");
    CHK(check_arch());
    const int E = c->embed_dim;
    const bool vitstr = c->arch == PARSEQ_ARCH_VITSTR;
    if (c->arch != PARSEQ_ARCH_PARSEQ && !vitstr) return fail(PARSEQ_E_INVALID, "arch=%d", c->arch);
    if (!vitstr && c->dec_depth != 1) return fail(PARSEQ_E_INVALID, "dec_depth=%d: only the reference's dec_depth == 1 is supported", c->dec_depth);
    if (E != 192 && E != 384 && E != 768) return fail(PARSEQ_E_INVALID, "embed_dim=%d not in {192, 384, 768}", E);
    if (c->enc_heads <= 0 || E / c->enc_heads != ATT_HD || E % c->enc_heads) return fail(PARSEQ_E_INVALID, "encoder head_dim must be 64 (embed_dim %d / heads %d)", E, c->enc_heads);
    if (!vitstr && (c->dec_heads <= 0 || E / c->dec_heads != 32 || E % c->dec_heads)) return fail(PARSEQ_E_INVALID, "decoder head_dim must be 32 (embed_dim %d / heads %d)", E, c->dec_heads);
    if (c->patch_h <= 0 || c->patch_w <= 0 || c->img_h % c->patch_h || c->img_w % c->patch_w || c->patch_w % 8)
        return fail(PARSEQ_E_INVALID, "unsupported image/patch geometry %dx%d / %dx%d", c->img_h, c->img_w, c->patch_h, c->patch_w);
    const int patch_tokens = (c->img_h / c->patch_h) * (c->img_w / c->patch_w);
    const int tokens = patch_tokens + (vitstr ? 1 : 0);
    // 128 tokens (32x128 crops, 4x8 patches) run the tuned attention kernels; any other count up to ATTG_THREADS (the 196 of
    // parseq-patch16-224, the 129 of ViTSTR) the token-count-generic ones
    if (tokens < 1 || tokens > ATTG_THREADS) return fail(PARSEQ_E_INVALID, "%d encoder tokens: supported range is [1, %d]", tokens, ATTG_THREADS);
    if (c->max_label_length < 1 || c->max_label_length + 1 > DEC_MAXL) return fail(PARSEQ_E_INVALID, "max_label_length=%d outside [1, %d]", c->max_label_length, DEC_MAXL - 1);
    if (vitstr && c->max_label_length + 2 > tokens) return fail(PARSEQ_E_INVALID, "max_label_length=%d needs %d tokens, the encoder has %d", c->max_label_length, c->max_label_length + 2, tokens);
    if (c->num_tokens < 3) return fail(PARSEQ_E_INVALID, "num_tokens=%d", c->num_tokens);

    auto* m = new parseq_model();
    m->cfg = *c;
    HIPCHK(hipGetDevice(&m->device));
    m->tokens = tokens;
    m->patch_tokens = patch_tokens;
    m->vitstr = vitstr;
    m->enc = vitstr ? "" : "encoder.";
    m->patch_k = 3 * c->patch_h * c->patch_w;
    m->classes = c->num_tokens - 2;
    const int64_t F = (int64_t)E * c->enc_mlp_ratio, Fd = (int64_t)E * c->dec_mlp_ratio;
    const std::string& pe = m->enc;
    if (vitstr) add_param(m, "cls_token", E);                 // timm VisionTransformer key order (class token first)
    else add_param(m, "pos_queries", (int64_t)(c->max_label_length + 1) * E);
    add_param(m, pe + "pos_embed", (int64_t)tokens * E);
    add_param(m, pe + "patch_embed.proj.weight", (int64_t)E * m->patch_k);
    add_param(m, pe + "patch_embed.proj.bias", E);
    for (int i = 0; i < c->enc_depth; ++i) {
        const std::string p = pe + "blocks." + std::to_string(i) + ".";
        add_param(m, p + "norm1.weight", E); add_param(m, p + "norm1.bias", E);
        add_param(m, p + "attn.qkv.weight", (int64_t)3 * E * E); add_param(m, p + "attn.qkv.bias", 3 * E);
        add_param(m, p + "attn.proj.weight", (int64_t)E * E); add_param(m, p + "attn.proj.bias", E);
        add_param(m, p + "norm2.weight", E); add_param(m, p + "norm2.bias", E);
        add_param(m, p + "mlp.fc1.weight", F * E); add_param(m, p + "mlp.fc1.bias", F);
        add_param(m, p + "mlp.fc2.weight", E * F); add_param(m, p + "mlp.fc2.bias", E);
    }
    add_param(m, pe + "norm.weight", E); add_param(m, pe + "norm.bias", E);
    if (!vitstr) {
        const std::string p = "decoder.layers.0.";
        for (const char* a : {"self_attn.", "cross_attn."}) {
            add_param(m, p + a + "in_proj_weight", (int64_t)3 * E * E); add_param(m, p + a + "in_proj_bias", 3 * E);
            add_param(m, p + a + "out_proj.weight", (int64_t)E * E); add_param(m, p + a + "out_proj.bias", E);
        }
        add_param(m, p + "linear1.weight", Fd * E); add_param(m, p + "linear1.bias", Fd);
        add_param(m, p + "linear2.weight", E * Fd); add_param(m, p + "linear2.bias", E);
        for (const char* n : {"norm1.", "norm2.", "norm_q.", "norm_c."}) { add_param(m, p + n + "weight", E); add_param(m, p + n + "bias", E); }
        add_param(m, "decoder.norm.weight", E); add_param(m, "decoder.norm.bias", E);
    }
    add_param(m, "head.weight", (int64_t)m->classes * E); add_param(m, "head.bias", m->classes);
    if (!vitstr) add_param(m, "text_embed.embedding.weight", (int64_t)c->num_tokens * E);
    hipError_t e = hipMalloc(&m->master, m->master_elems * sizeof(float));
    if (e != hipSuccess) { delete m; return fail(PARSEQ_E_HIP, "hipMalloc(%zu) failed: %s", m->master_elems * sizeof(float), hipGetErrorString(e)); }
    *out = m;
    return 0;
}

extern "C" void parseq_model_destroy(parseq_model* m) {
    if (!m) return;
    DevGuard dg(m->device);
    if (m->master) (void)hipFree(m->master);
    if (m->out_chunks) (void)hipFree(m->out_chunks);
    if (m->train_blocks_dev) (void)hipFree(m->train_blocks_dev);
    if (m->shadow_tab_dev) (void)hipFree(m->shadow_tab_dev);
    for (hipEvent_t e : m->grad_events) (void)hipEventDestroy(e);
    for (auto& ts : m->train_sides) {
        for (hipEvent_t e : ts.ev) if (e) (void)hipEventDestroy(e);
        if (ts.side) (void)hipStreamDestroy(ts.side);
    }
    delete m;
}

extern "C" int parseq_model_set_param(parseq_model* m, const char* key, const float* device_ptr, int64_t numel, void* stream) {
    if (!m || !key || !device_ptr) return fail(PARSEQ_E_INVALID, "null argument");
    auto it = m->index.find(key);
    if (it == m->index.end()) return fail(PARSEQ_E_INVALID, "unknown parameter key '%s'", key);
    ParamSpec& s = m->params[it->second];
    if (s.numel != numel) return fail(PARSEQ_E_INVALID, "parameter '%s': expected %lld elements, got %lld", key, (long long)s.numel, (long long)numel);
    DevGuard dg(m->device);
    HIPCHK(hipMemcpyAsync(m->master + s.offset, device_ptr, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    s.set = true;
    m->version++;
    return 0;
}

extern "C" int parseq_model_num_params(const parseq_model* m) { return m ? (int)m->params.size() : 0; }
extern "C" int parseq_model_param_info(const parseq_model* m, int index, const char** key, int64_t* numel) {
    if (!m || index < 0 || index >= (int)m->params.size()) return fail(PARSEQ_E_INVALID, "bad parameter index");
    if (key) *key = m->params[index].key.c_str();
    if (numel) *numel = m->params[index].numel;
    return 0;
}

// -------------------------------------------------------------------------------------------------------------------

template <typename T>
static int build_tables(parseq_plan* p, hipStream_t s) {
    const parseq_model* m = p->m;
    const parseq_config& c = m->cfg;
    const int E = c.embed_dim, npos = c.max_label_length + 1, ntok = c.num_tokens;
    const Weights<T> W = weights_of<T>(p);
    const std::string d = "decoder.layers.0.";
    // content K/V table: norm_c(content(pos, tok)) @ Wkv_self^T + bkv
    {
        const int rows = npos * ntok;
        const dim3 grid((rows + 3) / 4), block(256);
        T* ln = reinterpret_cast<T*>(p->ctab_ln);
        const float* emb = m->p("text_embed.embedding.weight"); const float* pq_ = m->p("pos_queries");
        const float* nw = m->p(d + "norm_c.weight"); const float* nb = m->p(d + "norm_c.bias");
        switch (E) {
            case 192: hipLaunchKernelGGL((content_ln_kernel<T, 192>), grid, block, 0, s, emb, pq_, nw, nb, ln, npos, ntok, c.dec_ln_eps); break;
            case 384: hipLaunchKernelGGL((content_ln_kernel<T, 384>), grid, block, 0, s, emb, pq_, nw, nb, ln, npos, ntok, c.dec_ln_eps); break;
            default:  hipLaunchKernelGGL((content_ln_kernel<T, 768>), grid, block, 0, s, emb, pq_, nw, nb, ln, npos, ntok, c.dec_ln_eps); break;
        }
        HIPCHK(hipGetLastError());
        CHK((run_gemm<T>(s, ARowMajor<T>{ln, E}, W.w(d + "self_attn.in_proj_weight") + (size_t)E * E, E, rows, 2 * E, E,
                         epi_store<T>(rows, 2 * E, m->p(d + "self_attn.in_proj_bias") + E, reinterpret_cast<T*>(p->kvtab), 2 * E))));
    }
    // position-query table: (norm_q(pos_queries[i]) @ Wq_self^T + bq) / sqrt(hd)      (modules.py:90, functional.py q_scaled)
    {
        T* ln = reinterpret_cast<T*>(p->ctab_ln);
        CHK((run_layernorm<T>(s, m->p("pos_queries"), m->p(d + "norm_q.weight"), m->p(d + "norm_q.bias"), ln, nullptr, npos, E, c.dec_ln_eps)));
        const float scale = sqrtf(1.0f / (float)(E / c.dec_heads));
        CHK((run_gemm<T>(s, ARowMajor<T>{ln, E}, W.w(d + "self_attn.in_proj_weight"), E, npos, E, E,
                         epi_store<float>(npos, E, m->p(d + "self_attn.in_proj_bias"), p->qself, E, scale))));
    }
    // self-attention score table: every (query position, key position, key token, head) dot product
    {
        const size_t total = (size_t)npos * npos * ntok * (E / DEC_HD);
        hipLaunchKernelGGL((score_table_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p->qself,
                           reinterpret_cast<const T*>(p->kvtab), p->stab, npos, ntok, E);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// Element offsets of every encoder block's parameters (encoder_blocks.h EncBlockParams) and of the tail's — identical in the fp32
// master, the bf16 copy and the block-planar bf16x3 pack (all three lay the tensors out alike).
static int build_block_table(parseq_plan* p, hipStream_t s) {
    const parseq_model* m = p->m;
    p->blocks_host.resize(m->cfg.enc_depth);
    auto off = [&](const std::string& key) { return (unsigned)m->params[m->index.at(key)].offset; };
    for (int i = 0; i < m->cfg.enc_depth; ++i) {
        const std::string b = m->enc + "blocks." + std::to_string(i) + ".";
        EncBlockParams& e = p->blocks_host[i];
        e.ln1_w = off(b + "norm1.weight"); e.ln1_b = off(b + "norm1.bias");
        e.wqkv = off(b + "attn.qkv.weight"); e.bqkv = off(b + "attn.qkv.bias");
        e.wproj = off(b + "attn.proj.weight"); e.bproj = off(b + "attn.proj.bias");
        e.ln2_w = off(b + "norm2.weight"); e.ln2_b = off(b + "norm2.bias");
        e.w1 = off(b + "mlp.fc1.weight"); e.b1 = off(b + "mlp.fc1.bias");
        e.w2 = off(b + "mlp.fc2.weight"); e.b2 = off(b + "mlp.fc2.bias");
    }
    HIPCHK(hipMemcpyAsync(p->blocks_dev, p->blocks_host.data(), p->blocks_host.size() * sizeof(EncBlockParams), hipMemcpyHostToDevice, s));
    if (!m->vitstr) {
        const int E_ = m->cfg.embed_dim;
        p->enc_tail.norm_w = off(m->enc + "norm.weight"); p->enc_tail.norm_b = off(m->enc + "norm.bias");
        p->enc_tail.wkv = off("decoder.layers.0.cross_attn.in_proj_weight") + (unsigned)E_ * E_;
        p->enc_tail.bkv = off("decoder.layers.0.cross_attn.in_proj_bias") + (unsigned)E_;
        p->enc_tail.heads = m->cfg.dec_heads;
    }
    return 0;
}

static int pack_weights(parseq_plan* p, hipStream_t s) {
    const parseq_model* m = p->m;
    for (const auto& ps : m->params)
        if (!ps.set) return fail(PARSEQ_E_STATE, "parameter '%s' has not been set", ps.key.c_str());
    if (p->precision == PARSEQ_BF16) {
        const size_t n = m->master_elems;
        hipLaunchKernelGGL(cvt_f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, m->master, reinterpret_cast<bf16_t*>(p->wpack), n);
        HIPCHK(hipGetLastError());
        if (!m->vitstr) CHK(build_tables<bf16_t>(p, s));
        {   // parameter offsets of the encoder blocks for the one-launch encoder (encoder_blocks.h): element offsets, identical in
            // the fp32 master and in the bf16 copy (both lay the tensors out alike)
            CHK(build_block_table(p, s));
            auto off = [&](const std::string& key) { return (unsigned)m->params[m->index.at(key)].offset; };
            if (!m->vitstr) {
                const int E_ = m->cfg.embed_dim;
                // head of the one-launch encoder: pos_embed + patch-embed bias as one table
                p->wpe_off = off(m->enc + "patch_embed.proj.weight");
                const int rows_ = m->tokens;
                hipLaunchKernelGGL(add_rowvec_kernel, dim3((unsigned)(((size_t)rows_ * E_ + 255) / 256)), dim3(256), 0, s, m->p(m->enc + "pos_embed"),
                                   m->p(m->enc + "patch_embed.proj.bias"), p->posb, rows_, E_);
                HIPCHK(hipGetLastError());
            }
        }
        if (p->wstep[0]) {       // decoder weights in MFMA-fragment order for the fused AR step
            const int E = m->cfg.embed_dim, Fd = E * m->cfg.dec_mlp_ratio;
            const Weights<bf16_t> W = weights_of<bf16_t>(p);
            const std::string d = "decoder.layers.0.";
            struct { const bf16_t* w; int N, K; } src[6] = {
                {W.w(d + "self_attn.out_proj.weight"), E, E}, {W.w(d + "cross_attn.in_proj_weight"), E, E},
                {W.w(d + "cross_attn.out_proj.weight"), E, E}, {W.w(d + "linear1.weight"), Fd, E},
                {W.w(d + "linear2.weight"), E, Fd}, {W.w("head.weight"), m->classes, E}};
            for (int i = 0; i < 6; ++i) {
                const int tiles = (src[i].N + 15) / 16;
                const size_t pieces = (size_t)tiles * (src[i].K / 64) * 128;
                hipLaunchKernelGGL(frag_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, src[i].w, src[i].N, src[i].K,
                                   src[i].K, p->wstep[i], tiles);
                HIPCHK(hipGetLastError());
            }
            hipLaunchKernelGGL(dec_qfold_kernel<bf16_t>, dim3(1), dim3(512), 0, s, src[0].w, m->p(d + "cross_attn.in_proj_weight"), m->p(d + "cross_attn.in_proj_bias"),
                               m->p(d + "norm1.weight"), m->p(d + "norm1.bias"), m->p(d + "self_attn.out_proj.bias"), m->p("pos_queries"), E,
                               m->cfg.max_label_length + 1, p->qfold);
            HIPCHK(hipGetLastError());
        }
    } else {
        if (p->precision == PARSEQ_BF16X3) {
            const size_t n = m->master_elems;
            hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, m->master, reinterpret_cast<unsigned char*>(p->wpack), n);
            HIPCHK(hipGetLastError());
            CHK(build_block_table(p, s));      // the one-launch encoder of this precision (encoder_blocks_x3.h)
            if (!m->vitstr) {                  // ... and its head: pos_embed + patch-embed bias as one table, the weight's offset in the pack
                const int E_ = m->cfg.embed_dim, rows_ = m->tokens;
                p->wpe_off = (unsigned)m->params[m->index.at(m->enc + "patch_embed.proj.weight")].offset;
                hipLaunchKernelGGL(add_rowvec_kernel, dim3((unsigned)(((size_t)rows_ * E_ + 255) / 256)), dim3(256), 0, s, m->p(m->enc + "pos_embed"),
                                   m->p(m->enc + "patch_embed.proj.bias"), p->posb, rows_, E_);
                HIPCHK(hipGetLastError());
            }
            if (p->wstep[0]) {       // decoder weights as hi | lo fragment pairs for the fused AR step, from the fp32 master
                const int E = m->cfg.embed_dim, Fd = E * m->cfg.dec_mlp_ratio;
                const std::string d = "decoder.layers.0.";
                struct { const float* w; int N, K; } src[6] = {
                    {m->p(d + "self_attn.out_proj.weight"), E, E}, {m->p(d + "cross_attn.in_proj_weight"), E, E},
                    {m->p(d + "cross_attn.out_proj.weight"), E, E}, {m->p(d + "linear1.weight"), Fd, E},
                    {m->p(d + "linear2.weight"), E, Fd}, {m->p("head.weight"), m->classes, E}};
                for (int i = 0; i < 6; ++i) {
                    const int tiles = (src[i].N + 15) / 16;
                    const size_t pieces = (size_t)tiles * (src[i].K / 64) * 256;
                    hipLaunchKernelGGL(frag_pack_x3_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, src[i].w, src[i].N, src[i].K,
                                       src[i].K, p->wstep[i], tiles);
                    HIPCHK(hipGetLastError());
                }
                hipLaunchKernelGGL(dec_qfold_kernel<float>, dim3(1), dim3(512), 0, s, src[0].w, src[1].w, m->p(d + "cross_attn.in_proj_bias"),
                                   m->p(d + "norm1.weight"), m->p(d + "norm1.bias"), m->p(d + "self_attn.out_proj.bias"), m->p("pos_queries"), E,
                                   m->cfg.max_label_length + 1, p->qfold);
                HIPCHK(hipGetLastError());
            }
        }
        SplitScope ss(p->precision == PARSEQ_BF16X3);
        if (!m->vitstr) CHK(build_tables<float>(p, s));
    }
    const int npos = m->cfg.max_label_length + 1;
    hipLaunchKernelGGL(cloze_mask_kernel, dim3(npos), dim3(LDT), 0, s, p->cloze, npos, LDT);
    HIPCHK(hipGetLastError());
    p->packed_version = m->version;
    return 0;
}

extern "C" void parseq_plan_destroy(parseq_plan* p);
extern "C" int parseq_plan_create(parseq_model* m, int max_batch, int precision, void* stream, parseq_plan** out) {
    return parseq_plan_create_ex(m, max_batch, precision, stream, nullptr, nullptr, nullptr, out);
}
extern "C" int parseq_plan_create_ex(parseq_model* m, int max_batch, int precision, void* stream, parseq_alloc_fn alloc, parseq_release_fn release, void* user,
                                     parseq_plan** out) {
    if (!m || !out || max_batch <= 0) return fail(PARSEQ_E_INVALID, "bad argument");
    if ((alloc == nullptr) != (release == nullptr)) return fail(PARSEQ_E_INVALID, "alloc and release go together");
    if (precision != PARSEQ_F32 && precision != PARSEQ_BF16 && precision != PARSEQ_BF16X3) return fail(PARSEQ_E_INVALID, "precision %d", precision);
    DevGuard dg(m->device);
    const parseq_config& c = m->cfg;
    const size_t E = c.embed_dim, N = m->tokens, B = max_batch, ts = precision == PARSEQ_BF16 ? 2 : 4;
    const size_t npos = c.max_label_length + 1, F = E * c.enc_mlp_ratio, Fd = E * c.dec_mlp_ratio;
    const size_t rows = B * N, drows = B * npos;
    auto* p = new parseq_plan();
    p->m = m; p->max_batch = max_batch; p->precision = precision;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0) p->num_cus = prop.multiProcessorCount; }
    size_t off = 0;
    const size_t o_wpack = carve(off, precision == PARSEQ_BF16 ? m->master_elems * 2 : (precision == PARSEQ_BF16X3 ? m->master_elems * 4 : 0));
    // fused AR step (decoder_step.h): bf16 fragment packs, or hi | lo pairs of them in the bf16x3 arithmetic (twice the elements)
    const bool step_ok = !m->vitstr && (precision == PARSEQ_BF16 || precision == PARSEQ_BF16X3) && E <= 384 && E % 64 == 0 && c.dec_mlp_ratio == 4;
    const size_t step_planes = precision == PARSEQ_BF16X3 ? 2 : 1;
    const size_t step_elems[6] = {frag_pack_elems(E, E), frag_pack_elems(E, E), frag_pack_elems(E, E), frag_pack_elems(Fd, E),
                                  frag_pack_elems(E, Fd), frag_pack_elems(m->classes, E)};
    size_t o_wstep[6];
    for (int i = 0; i < 6; ++i) o_wstep[i] = carve(off, step_ok ? step_elems[i] * 2 * step_planes : 0);
    const size_t o_kvtab = carve(off, npos * c.num_tokens * 2 * E * ts);
    const size_t o_qself = carve(off, npos * E * 4);
    const size_t o_ctab = carve(off, npos * c.num_tokens * E * ts);
    const size_t o_x = carve(off, rows * E * 4);
    const size_t o_xn = carve(off, rows * E * ts);
    const size_t o_q = carve(off, rows * E * ts), o_k = carve(off, rows * E * ts), o_vt = carve(off, rows * E * ts);
    const size_t o_ao = carve(off, rows * E * ts);
    const size_t o_h = carve(off, rows * F * ts);
    const size_t o_kmem = carve(off, rows * E * ts), o_vtmem = carve(off, rows * E * ts);
    const size_t o_stab = carve(off, npos * npos * c.num_tokens * (E / 32) * 4);
    const size_t o_sa = carve(off, drows * E * ts), o_tn = carve(off, drows * E * ts), o_ca = carve(off, drows * E * ts);
    // (the fused AR step keeps its linear2 partial sums here: ds_split workgroups per row tile x [max_batch][E] f32 — more than drows * Fd * ts when max_label_length is 1)
    const size_t o_hdn = carve(off, std::max<size_t>(drows * Fd * ts, (size_t)ds_split<384>() * B * E * 4));
    const size_t o_t = carve(off, drows * E * 4), o_qc = carve(off, drows * E * 4);
    const size_t o_tok = carve(off, B * LDT * 4), o_kpm = carve(off, B * LDT), o_eos = carve(off, B);
    const size_t o_cloze = carve(off, npos * LDT), o_qmu = carve(off, npos * LDT), o_cnt = carve(off, 64);
    const size_t o_blocks = carve(off, (size_t)c.enc_depth * sizeof(EncBlockParams));
    const size_t o_posb = carve(off, N * E * 4);
    const size_t o_qfold = carve(off, (3 * E + npos) * 4);
    p->arena_bytes = off;
    if (alloc) {
        p->arena = static_cast<unsigned char*>(alloc(off, user));
        if (!p->arena) { delete p; return fail(PARSEQ_E_HIP, "the caller's allocator returned NULL for the plan workspace (%zu bytes)", off); }
        if (reinterpret_cast<uintptr_t>(p->arena) % 256) { release(p->arena, user); delete p; return fail(PARSEQ_E_INVALID, "the caller's allocator returned a block that is not aligned to 256 bytes"); }
        p->arena_release = release; p->arena_user = user;
    } else {
        hipError_t e = hipMalloc(&p->arena, off);
        if (e != hipSuccess) { delete p; return fail(PARSEQ_E_HIP, "hipMalloc(%zu) for the plan workspace failed: %s", off, hipGetErrorString(e)); }
    }
    unsigned char* a = p->arena;
    if (step_ok) for (int i = 0; i < 6; ++i) p->wstep[i] = reinterpret_cast<bf16_t*>(a + o_wstep[i]);
    p->wpack = a + o_wpack; p->kvtab = a + o_kvtab; p->qself = (float*)(a + o_qself); p->ctab_ln = a + o_ctab;
    p->x = (float*)(a + o_x); p->xn = a + o_xn; p->q = a + o_q; p->k = a + o_k; p->vt = a + o_vt; p->ao = a + o_ao; p->h = a + o_h;
    p->kv_plane_elems = rows * E;      // elements of K (or V) at max_batch: [B][H][tokens][32]
    p->kmem = a + o_kmem; p->vmem = a + o_vtmem; p->stab = (float*)(a + o_stab); p->sa = a + o_sa; p->tn = a + o_tn; p->ca = a + o_ca; p->hdn = a + o_hdn;
    p->t = (float*)(a + o_t); p->qc = (float*)(a + o_qc);
    p->blocks_dev = reinterpret_cast<EncBlockParams*>(a + o_blocks);
    p->posb = reinterpret_cast<float*>(a + o_posb);
    p->qfold = reinterpret_cast<float*>(a + o_qfold);
    p->tok = (int*)(a + o_tok); p->kpm = a + o_kpm; p->eos_seen = a + o_eos; p->cloze = a + o_cloze; p->qmask_user = a + o_qmu; p->counters = (int*)(a + o_cnt);
    int r = pack_weights(p, (hipStream_t)stream);
    if (r != 0) { parseq_plan_destroy(p); return r; }
    *out = p;
    return 0;
}

extern "C" int parseq_plan_refresh(parseq_plan* p, void* stream) {
    if (!p) return fail(PARSEQ_E_INVALID, "null plan");
    DevGuard dg(p->m->device);
    return pack_weights(p, (hipStream_t)stream);
}

extern "C" void parseq_plan_destroy(parseq_plan* p) {
    if (!p) return;
    DevGuard dg(p->m->device);
    if (p->arena) { if (p->arena_release) p->arena_release(p->arena, p->arena_user); else (void)hipFree(p->arena); }
    delete p;
}

extern "C" size_t parseq_plan_workspace_bytes(const parseq_plan* p) { return p ? p->arena_bytes : 0; }

extern "C" int parseq_plan_set_profiling(parseq_plan* p, int enable) {
    if (!p) return fail(PARSEQ_E_INVALID, "null plan");
    p->prof.reset();
    p->prof.enabled = enable != 0;
    return 0;
}

extern "C" int parseq_plan_get_profile(parseq_plan* p, int index, const char** name, double* total_ms, int64_t* launches) {
    if (!p) return fail(PARSEQ_E_INVALID, "null plan");
    if (index < 0 || index >= T_COUNT) return 1;          // past the end (not an error: lets the caller iterate)
    p->prof.collect();
    if (name) *name = kProfNames[index];
    if (total_ms) *total_ms = p->prof.total_ms[index];
    if (launches) *launches = p->prof.launches[index];
    return 0;
}

// -------------------------------------------------------------------------------------------------------------------
// encoder
