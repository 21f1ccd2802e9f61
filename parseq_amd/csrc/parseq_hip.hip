// libparseq_hip.so — C ABI (include/parseq_hip.h) and launch orchestration of the PARSeq inference path on gfx950.
// Host side only decides WHICH kernels run in WHAT order on the caller's stream; all arithmetic is in the kernels of
// gemm.h / encoder_attn.h / decoder_attn.h / rowops.h.  No CPU fallback exists: without a gfx950 device every entry
// point fails with PARSEQ_E_ARCH / PARSEQ_E_HIP.
#include "../../include/parseq_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "decoder_attn.h"
#include "decoder_step.h"
#include "resize.h"
#include "encoder_attn.h"
#include "encoder_panel.h"
#include "encoder_mlp.h"
#include "encoder_attn_fused.h"
#include "encoder_blocks.h"
#include "encoder_blocks_x3.h"
#include "gemm.h"
#include "rowops.h"
#include "train_ops.h"

using namespace pq;

// -------------------------------------------------------------------------------------------------------------------
// errors
// -------------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(PARSEQ_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)                   \
    do {                            \
        int r_ = (expr);            \
        if (r_ != 0) return r_;     \
    } while (0)

// Every entry point that takes a model or a plan runs on THAT object's device, whatever device is current in the calling
// thread (a model moved to cuda:1 while cuda:0 is current must not launch on device 0 against device-1 pointers); the
// caller's current device is restored on return.  The stream passed in must belong to the object's device.
struct DevGuard {
    int prev = -1;
    bool switched = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev && hipSetDevice(dev) == hipSuccess) switched = true;
    }
    ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};

// precision bf16x3: the storage type of every activation is float (all kernels of the exact-f32 path are shared), but the
// GEMMs and the 128-token encoder attention evaluate their products on bf16 pairs (gemm.h SPLIT, attn_split_kernel) and the
// weights are read from the plan's block-planar hi / lo copy.  The mode of the call in progress on this host thread:
static thread_local bool g_split = false;
struct SplitScope {
    bool prev;
    explicit SplitScope(bool on) : prev(g_split) { g_split = on; }
    ~SplitScope() { g_split = prev; }
};

// -------------------------------------------------------------------------------------------------------------------
// optional per-kernel-family timing with HIP events on the caller's stream (bench.py's roofline leg)
// -------------------------------------------------------------------------------------------------------------------
enum ProfTag { T_PATCH, T_LN, T_QKV, T_ATTN, T_PROJ, T_FC1, T_FC2, T_MLP, T_ATTNF, T_BLOCKS, T_KVMEM, T_DEC_SA, T_DEC_GEMM, T_DEC_CA, T_DEC_LN, T_DEC_MISC, T_DEC_PRE, T_DEC_POST, T_COUNT };
static const char* const kProfNames[T_COUNT] = {"enc.patch_embed_gemm", "enc.layernorm", "enc.qkv_gemm", "enc.attention", "enc.proj_gemm",
                                                "enc.fc1_gelu_gemm", "enc.fc2_gemm", "enc.mlp_fused", "enc.attn_fused", "enc.blocks_fused", "dec.memory_kv_gemm", "dec.self_attention", "dec.gemm",
                                                "dec.cross_attention", "dec.layernorm", "dec.misc", "dec.step_pre", "dec.step_post"};
struct Profiler {
    bool enabled = false;
    std::vector<hipEvent_t> pool;          // events, used pairwise
    std::vector<int> tags;                 // tag of pair i
    size_t used = 0;                       // pairs in flight
    double total_ms[T_COUNT] = {0};
    long long launches[T_COUNT] = {0};
    int begin(int tag, hipStream_t s) {
        if (!enabled) return -1;
        if ((used + 1) * 2 > pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
            pool.push_back(a); pool.push_back(b);
        }
        if (tags.size() <= used) tags.resize(used + 1);
        tags[used] = tag;
        (void)hipEventRecord(pool[2 * used], s);
        return (int)used++;
    }
    void end(int id, hipStream_t s) { if (id >= 0) (void)hipEventRecord(pool[2 * id + 1], s); }
    void collect() {
        for (size_t i = 0; i < used; ++i) {
            float ms = 0.f;
            if (hipEventSynchronize(pool[2 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms, pool[2 * i], pool[2 * i + 1]) == hipSuccess) {
                total_ms[tags[i]] += ms; launches[tags[i]]++;
            }
        }
        used = 0;
    }
    void reset() { collect(); for (int t = 0; t < T_COUNT; ++t) { total_ms[t] = 0; launches[t] = 0; } }
    ~Profiler() { for (auto e : pool) (void)hipEventDestroy(e); }
};
struct ProfScope {
    Profiler* p; int id; hipStream_t s;
    ProfScope(Profiler* p_, int tag, hipStream_t s_) : p(p_), id(p_ ? p_->begin(tag, s_) : -1), s(s_) {}
    ~ProfScope() { if (p) p->end(id, s); }
};

extern "C" int parseq_abi_version(void) { return PARSEQ_ABI_VERSION; }
extern "C" const char* parseq_last_error(void) { return g_err; }

// -------------------------------------------------------------------------------------------------------------------
// model
// -------------------------------------------------------------------------------------------------------------------
struct ParamSpec { std::string key; int64_t numel; size_t offset; bool set; };

struct parseq_model {
    parseq_config cfg;
    int device = 0;
    int tokens = 0;           // encoder sequence length per image (patch tokens + the class token of ViTSTR)
    int patch_tokens = 0;     // patch tokens per image
    bool vitstr = false;      // cfg.arch == PARSEQ_ARCH_VITSTR: class token + per-token head, no decoder
    std::string enc;          // key prefix of the encoder parameters: "encoder." (PARSeq) or "" (ViTSTR)
    int patch_k = 0;          // 3 * patch_h * patch_w
    int classes = 0;          // num_tokens - 2
    int train_precision = PARSEQ_F32;     // training step: PARSEQ_F32 (exact products) or PARSEQ_BF16 (GEMM operands rounded to bf16)
    std::vector<ParamSpec> params;
    std::unordered_map<std::string, int> index;
    float* master = nullptr;  // device, all parameters fp32 back to back (each 16-byte aligned)
    size_t master_elems = 0;
    uint64_t version = 0;
    // parseq_model_get_params: the caller's destination pointers of the last call and the device table of copy pieces built from them
    std::vector<float*> out_ptrs;
    void* out_chunks = nullptr;
    int out_chunk_count = 0;

    const float* p(const std::string& key) const { return master + params[index.at(key)].offset; }
};

static void add_param(parseq_model* m, const std::string& key, int64_t numel) {
    ParamSpec s{key, numel, m->master_elems, false};
    m->index[key] = (int)m->params.size();
    m->params.push_back(s);
    m->master_elems += (size_t)((numel + 31) / 32 * 32);   // every tensor starts on a 32-element boundary: 16-byte aligned in bf16,
                                                             // and whole 32-element blocks of the bf16x3 hi / lo layout (gemm.h)
}

static int check_arch() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(PARSEQ_E_HIP, "hipGetDevice failed: %s (no ROCm device visible?)", hipGetErrorString(e));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PARSEQ_E_ARCH, "libparseq_hip is built for gfx950 (MI355X) only; device %d is %s", dev, prop.gcnArchName);
    return 0;
}

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
// plan
// -------------------------------------------------------------------------------------------------------------------
__global__ void cvt_f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        const float o[4] = {v.x, v.y, v.z, v.w};
        store4<bf16_t>(dst + i, o);
    } else {
        for (size_t j = i; j < n; ++j) dst[j] = static_cast<bf16_t>(src[j]);
    }
}

// bf16x3 weights (gemm.h SPLIT): flat block-planar copy of the fp32 master — elements [32 b, 32 b + 32) -> bytes [128 b, 128 b + 64)
// hi = bf16(v), bytes [128 b + 64, 128 b + 128) lo = bf16(v - hi).  Every tensor starts on a 32-element boundary and every GEMM
// weight row is a multiple of 32 long, so blocks never straddle rows and element offsets into the copy equal those into the master.
__global__ void split_pack_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;       // 4 consecutive elements
    if (i >= n) return;
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    uint2 hi, lo;
    split4(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, hi, lo);
    unsigned char* d = dst + (i >> 5) * 128 + (i & 31) * 2;
    *reinterpret_cast<uint2*>(d) = hi;
    *reinterpret_cast<uint2*>(d + 64) = lo;
}

__global__ void cloze_mask_kernel(unsigned char* __restrict__ mask, int n, int ld) {
    // model.py:117,157: causal triu(1) with triu(2) cleared -> query i may not see key i + 1 only
    const int i = blockIdx.x, j = threadIdx.x;
    if (i < n && j < ld) mask[i * ld + j] = (j == i + 1) ? 1 : 0;
}

// ViTSTR sequence assembly: x[b][0] = cls_token + pos_embed[0]; x[b][1 + t] = xp[b][t] (patch rows, pos_embed already added).
__global__ void insert_cls_kernel(const float* __restrict__ xp, const float* __restrict__ cls, const float* __restrict__ pos0,
                                  float* __restrict__ x, int B, int Np, int E) {
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t e4 = E / 4, per_img = (size_t)(Np + 1) * e4;
    if (i4 >= (size_t)B * per_img) return;
    const size_t b = i4 / per_img, r = i4 - b * per_img, t = r / e4, c4 = r - t * e4;
    float4 v;
    if (t == 0) {
        const float4 a = reinterpret_cast<const float4*>(cls)[c4], q = reinterpret_cast<const float4*>(pos0)[c4];
        v = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
    } else {
        v = reinterpret_cast<const float4*>(xp)[(b * Np + (t - 1)) * e4 + c4];
    }
    reinterpret_cast<float4*>(x)[i4] = v;
}

struct parseq_plan {
    parseq_model* m = nullptr;
    int max_batch = 0;
    int precision = PARSEQ_BF16;
    uint64_t packed_version = ~0ull;
    unsigned char* arena = nullptr;
    size_t arena_bytes = 0;
    // carved pointers (typed at use)
    void* wpack = nullptr;         // all parameters in storage type T (bf16 mode only; f32 mode aliases the master)
    bf16_t* wstep[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // fragment-packed decoder weights (decoder_step.h):
                                   // self out_proj, cross q-proj, cross out_proj, linear1, linear2, head; bf16 mode, E <= 384
    void* kvtab = nullptr;         // T [npos][num_tokens][2E]
    float* qself = nullptr;        // [npos][E], pre-scaled
    void* ctab_ln = nullptr;       // T [npos * num_tokens][E] scratch for table build
    float* x = nullptr;            // fp32 [B*N][E]
    void *xn = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *ao = nullptr, *h = nullptr;
    void *kmem = nullptr, *vmem = nullptr;   // cross-attention K and V of memory, head-split [B][H][N][32]
    float* stab = nullptr;         // [npos][npos][num_tokens][H] self-attention score table
    void *sa = nullptr, *tn = nullptr, *ca = nullptr, *hdn = nullptr;   // tn: unused since LayerNorm moved into the GEMM A-loaders
    float *t = nullptr, *qc = nullptr;
    int* tok = nullptr;            // [B][LDT]
    unsigned char* kpm = nullptr;  // [B][LDT]
    unsigned char* eos_seen = nullptr;
    unsigned char* cloze = nullptr;  // [npos][LDT]
    unsigned char* qmask_user = nullptr;  // [npos][LDT] staging for parseq_decode_logits
    int* counters = nullptr;       // [0] rows that have seen an EOS, [1] step at which the reference would have stopped (ar_len)
    int last_batch = 0;            // batch of the most recent parseq_encode (kvmem valid for it)
    int num_cus = 256;             // compute units of the device (tail-round avoidance of the one- and two-workgroup-per-CU kernels)
    bool fused_step = getenv("PARSEQ_NO_FUSED_STEP") == nullptr;   // diagnostics: fall back to the per-op AR step
    bool fused_attn = getenv("PARSEQ_NO_FUSED_ATTN") == nullptr;   // diagnostics: qkv panel GEMM + attention + proj GEMM instead of encoder_attn_fused.h
    bool mlp_resident = getenv("PARSEQ_MLP_RELOAD") == nullptr;    // diagnostics: the fused MLP's first form (x re-read by the epilogue)
    bool fused_blocks = getenv("PARSEQ_NO_FUSED_BLOCKS") == nullptr;   // diagnostics: one launch per branch instead of encoder_blocks.h
    bool fused_x3 = getenv("PARSEQ_NO_FUSED_X3") == nullptr;
         // diagnostics: bf16x3 encoder through the per-op kernels instead of encoder_blocks_x3.h
    EncBlockParams* blocks_dev = nullptr;                           // [enc_depth] parameter pointers of encoder_blocks.h (bf16 mode)
    std::vector<EncBlockParams> blocks_host;                        // source of the asynchronous upload (must outlive it)
    EncTailParams enc_tail{0, 0, 0, 0, nullptr, nullptr, 0};        // final norm + memory K / V projection inside the one-launch encoder (offsets; pointers filled per call)
    bool fused_tail = getenv("PARSEQ_NO_FUSED_TAIL") == nullptr;    // diagnostics: final LayerNorm and K / V GEMM as their own launches
    float* posb = nullptr;                                          // [tokens][E] pos_embed + patch-embed bias (the one-launch encoder's head)
    unsigned wpe_off = 0;                                           // element offset of patch_embed.proj.weight in the weight pack
    bool fused_head = getenv("PARSEQ_NO_FUSED_HEAD") == nullptr;    // diagnostics: patch embedding as its own launch
    Profiler prof;
};
constexpr int LDT = 32;            // row pitch of token / mask arrays

static size_t carve(size_t& off, size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
}

template <typename T> struct Weights {
    const parseq_model* m; const T* base;
    const T* w(const std::string& key) const { return base + m->params[m->index.at(key)].offset; }
};

template <typename T>
static Weights<T> weights_of(const parseq_plan* p) {
    if constexpr (sizeof(T) == 4) return Weights<T>{p->m, reinterpret_cast<const T*>(p->precision == PARSEQ_BF16X3 ? p->wpack : (void*)p->m->master)};
    else return Weights<T>{p->m, reinterpret_cast<const T*>(p->wpack)};
}

template <typename TO>
static int run_layernorm(hipStream_t s, const float* x, const float* w, const float* b, TO* out, float* out32, int rows, int E, float eps) {
    const dim3 grid((rows + 3) / 4), block(256);
    switch (E) {
        case 192: hipLaunchKernelGGL((layernorm_kernel<TO, 192>), grid, block, 0, s, x, w, b, out, out32, rows, eps); break;
        case 384: hipLaunchKernelGGL((layernorm_kernel<TO, 384>), grid, block, 0, s, x, w, b, out, out32, rows, eps); break;
        case 768: hipLaunchKernelGGL((layernorm_kernel<TO, 768>), grid, block, 0, s, x, w, b, out, out32, rows, eps); break;
        default: return fail(PARSEQ_E_INVALID, "layernorm: E=%d not in {192, 384, 768}", E);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int run_layernorm_split(hipStream_t s, const float* x, const float* w, const float* b, unsigned char* out, int rows, int E, float eps) {
    const dim3 grid((rows + 3) / 4), block(256);
    switch (E) {
        case 384: hipLaunchKernelGGL((layernorm_split_kernel<384>), grid, block, 0, s, x, w, b, out, rows, eps); break;
        case 768: hipLaunchKernelGGL((layernorm_split_kernel<768>), grid, block, 0, s, x, w, b, out, rows, eps); break;
        default: return fail(PARSEQ_E_INVALID, "split layernorm: E=%d not in {384, 768}", E);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// GEMM dispatch: big tiles for the encoder's M = batch * 128 rows, small tiles for the decoder's M = batch (* 26).
template <typename T, typename ALoad, typename Epi>
static int run_gemm(hipStream_t s, const ALoad& a, const T* W, int ldw, int M, int N, int K, const Epi& epi, bool force_small = false) {
    if constexpr (sizeof(T) == 4) {
        if (g_split) {      // bf16x3: W is the block-planar hi / lo copy, products are three bf16 MFMAs (gemm.h SPLIT)
            if (K % 32) return fail(PARSEQ_E_INVALID, "bf16x3 GEMM: K=%d is not a multiple of 32", K);
#ifndef PQ_X3_KB
#define PQ_X3_KB 128
#define PQ_X3_NBUF 2
#endif
            if (M >= 4096 && !force_small) HIPCHK((launch_gemm<T, 128, 128, 2, 2, PQ_X3_KB, PQ_X3_NBUF, ALoad, Epi, true>(s, a, W, ldw, M, N, K, epi)));
#ifndef PQ_X3_SMALL_BM
#define PQ_X3_SMALL_BM 32
#define PQ_X3_SMALL_KB 1536
#endif
            else HIPCHK((launch_gemm<T, PQ_X3_SMALL_BM, PQ_X3_SMALL_BM, 2, 2, PQ_X3_SMALL_KB, 1, ALoad, Epi, true>(s, a, W, ldw, M, N, K, epi)));
            return 0;
        }
    }
    if (M >= 4096 && !force_small) HIPCHK((launch_gemm<T, 128, 128, 2, 2, 128, 2>(s, a, W, ldw, M, N, K, epi)));
    else HIPCHK((launch_gemm<T, 64, 64, 2, 2, 768, 1>(s, a, W, ldw, M, N, K, epi)));
    return 0;
}

// out = epi(LayerNorm(x[M, E]; g, b, eps) W^T): the LayerNorm rides in the GEMM's A-operand loader (gemm.h ALayerNorm).  One
// exception, now for speed only: bf16x3 products with the 128 x 128 tile configuration (M >= 4096) run the LayerNorm as its own
// kernel into `scratch` ([M, E] f32) and the GEMM with the row-major loader — or, with -DPQ_X3_LN_STATS=1, a statistics-only pass
// and the ALayerNormStats loader; both fused forms measured no faster than the separate launch (the split GEMM is bound by its LDS
// staging pass, which the loader's arithmetic lengthens).  History: this combination used to give wrong values in rows 6, 7 mod 8 of
// a tile whenever two workgroups shared a compute unit; the cause was in the loader's packed-f32 arithmetic (gemm.h ln_apply4),
// not in the statistics prologue, and is fixed there — tools/x3_diag2.py is the reproducer, exact and deterministic since.
template <typename T, int E, typename Epi>
static int run_ln_gemm(hipStream_t s, const float* x, const float* g, const float* b, float eps, const T* W, int M, int N, const Epi& epi, void* scratch) {
    if constexpr (sizeof(T) == 4) {
        if (g_split && M >= 4096) {
            if (!scratch) return fail(PARSEQ_E_STATE, "run_ln_gemm: no LayerNorm scratch");
#ifndef PQ_X3_LN_STATS
#define PQ_X3_LN_STATS 0
#endif
            if (PQ_X3_LN_STATS) {       // row statistics only (M x 2 floats); the GEMM's loader normalises from them
                hipLaunchKernelGGL((ln_stats_kernel<E>), dim3((M + 3) / 4), dim3(256), 0, s, x, reinterpret_cast<float*>(scratch), M, eps);
                HIPCHK(hipGetLastError());
                return run_gemm<T>(s, ALayerNormStats<T, E>{x, g, b, reinterpret_cast<const float*>(scratch)}, W, E, M, N, E, epi);
            }
            CHK((run_layernorm<float>(s, x, g, b, reinterpret_cast<float*>(scratch), nullptr, M, E, eps)));
            return run_gemm<T>(s, ARowMajor<T>{reinterpret_cast<const T*>(scratch), E}, W, E, M, N, E, epi);
        }
    }
    return run_gemm<T>(s, ALayerNorm<T, E>{x, g, b, eps, 0, nullptr}, W, E, M, N, E, epi);
}

// run_ln_gemm with the embedding width chosen at run time (the encoder's per-op path)
template <typename T, typename Epi>
static int run_ln_gemm_e(hipStream_t s, int E, const float* x, const float* g, const float* b, float eps, const T* W, int M, int N, const Epi& epi, void* scratch) {
    switch (E) {
        case 192: return run_ln_gemm<T, 192>(s, x, g, b, eps, W, M, N, epi, scratch);
        case 384: return run_ln_gemm<T, 384>(s, x, g, b, eps, W, M, N, epi, scratch);
        case 768: return run_ln_gemm<T, 768>(s, x, g, b, eps, W, M, N, epi, scratch);
        default: return fail(PARSEQ_E_INVALID, "LayerNorm-fused GEMM: E=%d not in {192, 384, 768}", E);
    }
}

static EpiBase epi_base(int M, int N, const float* bias) { EpiBase b; b.M = M; b.N = N; b.bias = bias; return b; }
template <typename TO> static EpiStore<TO> epi_store(int M, int N, const float* bias, TO* out, int ldo, float scale = 1.f, int period = 0, int stride = 0, int offset = 0) {
    EpiStore<TO> e; static_cast<EpiBase&>(e) = epi_base(M, N, bias); e.out = out; e.ldo = ldo; e.period = period; e.stride = stride; e.offset = offset; e.scale = scale; return e;
}
template <typename TO> static EpiGelu<TO> epi_gelu(int M, int N, const float* bias, TO* out, int ldo) {
    EpiGelu<TO> e; static_cast<EpiBase&>(e) = epi_base(M, N, bias); e.out = out; e.ldo = ldo; return e;
}
static EpiResid epi_resid(int M, int N, const float* bias, float* x, int ldx) {
    EpiResid e; static_cast<EpiBase&>(e) = epi_base(M, N, bias); e.x = x; e.ldx = ldx; return e;
}
static EpiAddTable epi_table(int M, int N, const float* bias, float* x, int ldx, const float* table, int ldt, int period, int offset) {
    EpiAddTable e; static_cast<EpiBase&>(e) = epi_base(M, N, bias); e.x = x; e.ldx = ldx; e.table = table; e.ldt = ldt; e.period = period; e.offset = offset; return e;
}

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
        }
    } else {
        if (p->precision == PARSEQ_BF16X3) {
            const size_t n = m->master_elems;
            hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, m->master, reinterpret_cast<unsigned char*>(p->wpack), n);
            HIPCHK(hipGetLastError());
            CHK(build_block_table(p, s));      // the one-launch encoder of this precision (encoder_blocks_x3.h)
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
    if (!m || !out || max_batch <= 0) return fail(PARSEQ_E_INVALID, "bad argument");
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
    const size_t o_hdn = carve(off, drows * Fd * ts);
    const size_t o_t = carve(off, drows * E * 4), o_qc = carve(off, drows * E * 4);
    const size_t o_tok = carve(off, B * LDT * 4), o_kpm = carve(off, B * LDT), o_eos = carve(off, B);
    const size_t o_cloze = carve(off, npos * LDT), o_qmu = carve(off, npos * LDT), o_cnt = carve(off, 64);
    const size_t o_blocks = carve(off, (size_t)c.enc_depth * sizeof(EncBlockParams));
    const size_t o_posb = carve(off, N * E * 4);
    p->arena_bytes = off;
    hipError_t e = hipMalloc(&p->arena, off);
    if (e != hipSuccess) { delete p; return fail(PARSEQ_E_HIP, "hipMalloc(%zu) for the plan workspace failed: %s", off, hipGetErrorString(e)); }
    unsigned char* a = p->arena;
    if (step_ok) for (int i = 0; i < 6; ++i) p->wstep[i] = reinterpret_cast<bf16_t*>(a + o_wstep[i]);
    p->wpack = a + o_wpack; p->kvtab = a + o_kvtab; p->qself = (float*)(a + o_qself); p->ctab_ln = a + o_ctab;
    p->x = (float*)(a + o_x); p->xn = a + o_xn; p->q = a + o_q; p->k = a + o_k; p->vt = a + o_vt; p->ao = a + o_ao; p->h = a + o_h;
    p->kmem = a + o_kmem; p->vmem = a + o_vtmem; p->stab = (float*)(a + o_stab); p->sa = a + o_sa; p->tn = a + o_tn; p->ca = a + o_ca; p->hdn = a + o_hdn;
    p->t = (float*)(a + o_t); p->qc = (float*)(a + o_qc);
    p->blocks_dev = reinterpret_cast<EncBlockParams*>(a + o_blocks);
    p->posb = reinterpret_cast<float*>(a + o_posb);
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
    if (p->arena) (void)hipFree(p->arena);
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
// -------------------------------------------------------------------------------------------------------------------
template <typename T>
static int run_enc_attention(hipStream_t s, const T* q, const T* k, const T* vt, T* ao, int bh, int heads, bool v_rowmajor = false, int tokens = ATT_N, bool split_out = false) {
    const float scale = 1.0f / sqrtf((float)ATT_HD);
    if (tokens != ATT_N) {
        if (!v_rowmajor) return fail(PARSEQ_E_INVALID, "token-count-generic attention expects row-major V");
        if constexpr (sizeof(T) == 2) {
            // bf16: the MFMA kernel padded to a multiple of 32 keys, one wave per 32 queries (ViTSTR: 5 waves, patch16-224: 7)
            const int nt32 = (tokens + 31) / 32;
#define PQ_ATTN_N(NT)                                                                                                                   \
            if (nt32 == NT) {                                                                                                           \
                static LdsAttr attr_;                                                                                                   \
                HIPCHK(attr_.ensure(reinterpret_cast<const void*>(attn_mfma_n_kernel<NT>), attn_mfma_n_lds<NT>()));                     \
                hipLaunchKernelGGL((attn_mfma_n_kernel<NT>), dim3(bh), dim3(64 * NT), attn_mfma_n_lds<NT>(), s, q, k, vt, ao, heads, tokens, scale); \
                HIPCHK(hipGetLastError());                                                                                              \
                return 0;                                                                                                               \
            }
            PQ_ATTN_N(1) PQ_ATTN_N(2) PQ_ATTN_N(3) PQ_ATTN_N(4) PQ_ATTN_N(5) PQ_ATTN_N(6) PQ_ATTN_N(7) PQ_ATTN_N(8)
#undef PQ_ATTN_N
        }
        const size_t lds = (size_t)2 * tokens * ATT_HD * sizeof(float);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_generic_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // size varies per call
        hipLaunchKernelGGL((attn_generic_kernel<T>), dim3(bh), dim3(ATTG_THREADS), lds, s, q, k, vt, ao, heads, tokens, scale);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if constexpr (sizeof(T) == 2) {
        if (v_rowmajor) hipLaunchKernelGGL(attn_mfma_kernel<true>, dim3(bh), dim3(256), 0, s, q, k, vt, ao, heads, scale);
        else hipLaunchKernelGGL(attn_mfma_kernel<false>, dim3(bh), dim3(256), 0, s, q, k, vt, ao, heads, scale);
    } else if (g_split) {
        if (v_rowmajor) return fail(PARSEQ_E_INVALID, "bf16x3 attention expects V^T");
        if (split_out) {
            static LdsAttr attr_s;
            HIPCHK(attr_s.ensure(reinterpret_cast<const void*>(attn_split_kernel<true>), attn_split_lds()));
            hipLaunchKernelGGL(attn_split_kernel<true>, dim3(bh), dim3(256), attn_split_lds(), s, q, k, vt, ao, heads, scale);
        } else {
            static LdsAttr attr;
            HIPCHK(attr.ensure(reinterpret_cast<const void*>(attn_split_kernel<false>), attn_split_lds()));
            hipLaunchKernelGGL(attn_split_kernel<false>, dim3(bh), dim3(256), attn_split_lds(), s, q, k, vt, ao, heads, scale);
        }
    } else {
        if (v_rowmajor) return fail(PARSEQ_E_INVALID, "f32 attention expects V^T");
        constexpr size_t lds = (size_t)2 * ATT_N * ATT_HD * sizeof(float);
        static LdsAttr attr;
        HIPCHK(attr.ensure(reinterpret_cast<const void*>(attn_f32_kernel), lds));
        hipLaunchKernelGGL(attn_f32_kernel, dim3(bh), dim3(128), lds, s, q, k, vt, ao, heads, scale);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

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
    if (head_in_launch) {
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
        // K / V projection of it — in one launch with x resident in registers (encoder_blocks_x3.h); the MLP hidden buffer (idle on this
        // path) is the launch's per-image scratch (the parked residual stream and the attention output, 384 KiB per image)
        if (g_split && p->fused_x3 && p->fused_blocks && !m->vitstr && E == 384 && c.enc_mlp_ratio == 4 && N == ATT_N && M % 128 == 0) {
            const bool tail = p->fused_tail && memory_out == nullptr && c.dec_heads * DEC_HD == E;
            x3::EncTailX3 et{p->enc_tail.norm_w, p->enc_tail.norm_b, p->enc_tail.wkv, p->enc_tail.bkv, nullptr, nullptr, p->enc_tail.heads};
            if (tail) { et.kmem = reinterpret_cast<float*>(p->kmem); et.vmem = reinterpret_cast<float*>(p->vmem); }
            {
                ProfScope ps_(&p->prof, T_BLOCKS, s);
                HIPCHK((x3::launch_enc_blocks_x3<384>(s, p->x, p->wpack, m->master_elems * sizeof(float), m->master, p->blocks_dev, c.enc_depth,
                                                      c.enc_ln_eps, M, reinterpret_cast<float*>(p->h), et)));
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

static int check_call(parseq_plan* p, int batch, int images_dtype) {
    if (!p) return fail(PARSEQ_E_INVALID, "null plan");
    if (batch <= 0 || batch > p->max_batch) return fail(PARSEQ_E_INVALID, "batch %d outside (0, %d]", batch, p->max_batch);
    if (images_dtype != PARSEQ_F32 && images_dtype != PARSEQ_BF16 && images_dtype != PARSEQ_U8) return fail(PARSEQ_E_INVALID, "images_dtype %d", images_dtype);
    if (p->packed_version != p->m->version) return fail(PARSEQ_E_STATE, "model parameters changed after the plan was packed; call parseq_plan_refresh");
    return 0;
}

static int encode_dispatch(parseq_plan* p, const void* images, int images_dtype, int batch, float* memory_out, hipStream_t s) {
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

// -------------------------------------------------------------------------------------------------------------------
// decoder
// -------------------------------------------------------------------------------------------------------------------
// Cross-attention of Lq queries per image against the plan's cached memory K / V: tuned kernels for 128 memory tokens
// (streaming AR kernel, MFMA multi-query kernel), the key-count-generic kernel otherwise.
template <typename T, int E>
static int run_cross_attention(parseq_plan* p, hipStream_t s, int B, int Lq, float scale, T* ca) {
    const int H = p->m->cfg.dec_heads, NK = p->m->tokens;
    const T* kmem = reinterpret_cast<const T*>(p->kmem); const T* vmem = reinterpret_cast<const T*>(p->vmem);
    const float* qc_ = p->qc;
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
        hipLaunchKernelGGL((dec_cross_attn_ar_kernel<T, E>), dim3(B), dim3(E), 0, s, qc_, kmem, vmem, scale, ca);
    } else if constexpr (sizeof(T) == 2) {
        hipLaunchKernelGGL(dec_cross_attn_multi_mfma_kernel, dim3((B * H + 3) / 4), dim3(256), 0, s, qc_, kmem, vmem, H, Lq, scale, ca, B * H);
    } else {
        if (g_split)      // bf16x3: the matrix-core kernel on bf16 pairs
            hipLaunchKernelGGL(dec_cross_attn_multi_mfma_x3_kernel, dim3((B * H + 1) / 2), dim3(128), 0, s, qc_, kmem, vmem, H, Lq, scale, ca, B * H);
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
static int ar_loop_fused(parseq_plan* p, hipStream_t s, int B, int num_steps, float* logits, bool testing) {
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
    static LdsAttr attr_mid, attr_mlp;
    HIPCHK(attr_mid.ensure(reinterpret_cast<const void*>(dec_step_mid_kernel<E, X3>), dec_step_mid_lds<E, X3>()));
    HIPCHK(attr_mlp.ensure(reinterpret_cast<const void*>(dec_step_mlp_kernel<E, X3>), dec_step_mlp_lds<E, X3>()));
    for (int i = 0; i <= num_steps; ++i) {
        const int do_finish = i > 0, do_start = i < num_steps;
        // the pick of position i - 1 feeds step i: needed while there is a step to start
        const int argmax_mode = do_start ? (testing ? 2 : 1) : 0;
        {
            ProfScope ps_(&p->prof, T_DEC_PRE, s);
            hipLaunchKernelGGL((dec_step_mid_kernel<E, X3>), grid, block, (dec_step_mid_lds<E, X3>()), s, do_finish, do_start, i, M,
                               tq, partial, m->p(d + "linear2.bias"), m->p("decoder.norm.weight"), m->p("decoder.norm.bias"), c.dec_ln_eps,
                               p->wstep[5], m->p("head.bias"), C, logits, num_steps, argmax_mode, c.eos_id, eos_seen, eos_rows, ar_len,
                               p->stab, reinterpret_cast<const TS*>(p->kvtab), tok, LDT, c.num_tokens, npos, p->wstep[0],
                               m->p(d + "self_attn.out_proj.bias"), m->p("pos_queries"), m->p(d + "norm1.weight"), m->p(d + "norm1.bias"),
                               p->wstep[1], m->p(d + "cross_attn.in_proj_bias"), t, tq);
            HIPCHK(hipGetLastError());
        }
        if (!do_start) break;
        {
            ProfScope ps_(&p->prof, T_DEC_CA, s);
            CHK((run_cross_attention<TS, E>(p, s, B, 1, scale, ca)));
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
    const bool ar = flags & PARSEQ_FLAG_DECODE_AR, testing = flags & PARSEQ_FLAG_TESTING;
    int* eos_rows = p->counters; int* ar_len = p->counters + 1;
    hipLaunchKernelGGL(ar_init_kernel, dim3((B * LDT + 255) / 256), dim3(256), 0, s, p->tok, LDT, B, c.bos_id, c.pad_id, p->eos_seen, p->counters, 2, num_steps);
    HIPCHK(hipGetLastError());
    if (ar) {
        // model.py:119-147.  All num_steps steps are always run (no per-step host sync); the step at which the reference
        // would have stopped is recorded on the device and only truncates the returned view.
        bool done = false;
        if constexpr (sizeof(T) == 2) {
            if (p->wstep[0] && p->fused_step && C <= 128 && c.dec_mlp_ratio == 4) {
                if (c.embed_dim == 384) { CHK((ar_loop_fused<384>(p, s, B, num_steps, logits, testing))); done = true; }
                else if (c.embed_dim == 192) { CHK((ar_loop_fused<192>(p, s, B, num_steps, logits, testing))); done = true; }
            }
        } else {
            // bf16x3: the same fused step on bf16 pairs (f32 tables, f32 memory K / V); the fp32 mode keeps the per-op kernels
            if (p->precision == PARSEQ_BF16X3 && p->wstep[0] && p->fused_step && C <= 128 && c.dec_mlp_ratio == 4) {
                if (c.embed_dim == 384) { CHK((ar_loop_fused<384, true>(p, s, B, num_steps, logits, testing))); done = true; }
                else if (c.embed_dim == 192) { CHK((ar_loop_fused<192, true>(p, s, B, num_steps, logits, testing))); done = true; }
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

// -------------------------------------------------------------------------------------------------------------------
// training step, decoder side: loss of system.py:168-199 and its gradients, fp32
// -------------------------------------------------------------------------------------------------------------------
// Scratch shared by the split-K partials of the MFMA GEMM and the partial column sums; part of the caller's workspace.
constexpr size_t TRAIN_SCRATCH_FLOATS = (size_t)16 << 20;
struct TrainCtx {
    hipStream_t s;
    float* scratch;      // TRAIN_SCRATCH_FLOATS floats
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
static int lin_bwd16(const TrainCtx& cx, const bf16_t* x16, const bf16_t* Wt16, const float* dy, const bf16_t* dy16, float* dW, float* db, float* dx,
                     bf16_t* dx16, int M, int N, int K, const float* dx_gelu_pre = nullptr, const bf16_t* dx_gelu_pre16 = nullptr) {
    if (!dy && !dy16) return fail(PARSEQ_E_INVALID, "lin_bwd16: no gradient");
    GemmExt ew; ew.b16 = true; ew.a16 = dy == nullptr;
    CHK(sgemm(cx, dy ? dy : reinterpret_cast<const float*>(dy16), 1, N, reinterpret_cast<const float*>(x16), K, 1, nullptr, nullptr, 0, 0, dW, K, N, K, M, 1.f, true,
              db, nullptr, nullptr, nullptr, &ew));
    if (!dx && !dx16) return 0;
    GemmExt ex; ex.b16 = true; ex.a16 = dy16 != nullptr; ex.c16 = dx16; ex.gelu_pre16 = dx_gelu_pre16;
    return sgemm(cx, dy16 ? reinterpret_cast<const float*>(dy16) : dy, N, 1, reinterpret_cast<const float*>(Wt16), 1, N, nullptr, nullptr, 0, 0, dx, K, M, K, N,
                 1.f, false, nullptr, nullptr, dx_gelu_pre, nullptr, &ex);
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
    size_t d_a, d_b, d_c, d_h, pm, d_kvc, d_kvm, d_kvm_p, d_content, d_pq, d_qb, row_loss, tgt_all, losses, counts, scratch, total;
    int KP;                          // passes per batch
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
    o.d_kvm_p = o.KP > 1 ? take(P * MS * 2 * E) : o.d_kvm;      // each pass's own d K | d V of the memory, folded into d_kvm after the batch
    o.d_content = take(M * E); o.d_pq = take(L * E); o.d_qb = take(MP * E);
    o.row_loss = take(MP); o.tgt_all = take((size_t)K * M); o.losses = take(K + 1); o.counts = take(K + 1); o.scratch = take(TRAIN_SCRATCH_FLOATS);
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
    const TrainCtx cx{s, w + o.scratch, m->train_precision == PARSEQ_BF16};

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
    const bool ca_loop = KP > 1 && train_attn_is_dec_bf16(cx, ca, 32) && !getenv("PARSEQ_TRAIN_NO_PASS_LOOP");
    if (ca_loop) { ca.dk = d_kvm; ca.dv = d_kvm + E; }
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
    size_t patches, layer0, layer_stride, x_last, n, hact, d_x, d_a, d_h, dqkv, tmp, scratch, total;
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
    o.dqkv = take(MS * 3 * E); o.tmp = take(MS * E); o.scratch = take(TRAIN_SCRATCH_FLOATS);
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
    const TrainCtx cx{s, w + o.scratch, m->train_precision == PARSEQ_BF16};
    hipLaunchKernelGGL(patches_kernel, dim3(MS), dim3(256), 0, s, images, m->cfg.img_h, m->cfg.img_w, m->cfg.patch_h, m->cfg.patch_w, w + o.patches);
    HIPCHK(hipGetLastError());
    CHK(lin_fwd(cx, w + o.patches, P("patch_embed.proj.weight"), P("patch_embed.proj.bias"), P("pos_embed"), S, w + o.x(0), MS, E, PK));
    const size_t elems = (size_t)MS * F;
    const bool shadows = train_enc_shadows(m), only16 = train_enc_bf16_only(m);
    if (shadows) {
        // this step's weights as bf16, both ways round (the backward entry reads the transposes from the same workspace)
        const char* names[4] = {"attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"};
        const int wn[4] = {3 * E, E, F, E}, wk[4] = {E, E, E, F};
        for (int i = 0; i < m->cfg.enc_depth; ++i)
            for (int j = 0; j < 4; ++j) {
                const EncShadowW sw = enc_shadow_w(o, w, i, j, E, F);
                hipLaunchKernelGGL(weight_shadow_kernel, dim3(wk[j] / 32, wn[j] / 32), dim3(256), 0, s, P("blocks." + std::to_string(i) + "." + names[j]), wn[j], wk[j], sw.w, sw.wt);
            }
        HIPCHK(hipGetLastError());
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
    const TrainCtx cx{s, w + o.scratch, m->train_precision == PARSEQ_BF16};
    const bool shadows = train_enc_shadows(m), only16 = train_enc_bf16_only(m);
    bf16_t* d_x16 = shadows ? reinterpret_cast<bf16_t*>(w + o.d_x16) : nullptr;
    bf16_t* d_h16 = shadows ? reinterpret_cast<bf16_t*>(w + o.d_h16) : nullptr;
    CHK(ln_bwd(cx, w + o.x_last, P("norm.weight"), dmemory, nullptr, d_x, G("norm.weight"), G("norm.bias"), tmp, MS, E, eps, d_x16));
    for (int i = m->cfg.enc_depth - 1; i >= 0; --i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        float* x = w + o.x(i); float* qkv = x + o.qkv; float* ao = x + o.ao; float* x_mid = x + o.x_mid; float* hpre = x + o.hpre;
        if (shadows) {
            const bf16_t* n1 = reinterpret_cast<const bf16_t*>(x + o.n1); const bf16_t* n2 = reinterpret_cast<const bf16_t*>(x + o.n2);
            const bf16_t* ao16 = reinterpret_cast<const bf16_t*>(ao); const bf16_t* hact16 = reinterpret_cast<const bf16_t*>(x + o.hact_l);
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
    return lin_bwd(cx, w + o.patches, P("patch_embed.proj.weight"), d_x, G("patch_embed.proj.weight"), G("patch_embed.proj.bias"), nullptr, MS, E, PK);
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
extern "C" int parseq_op_enc_blocks_x3(float* x, const float* master, const void* pack, int64_t master_elems, const uint32_t* offsets, int depth,
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
    HIPCHK((x3::launch_enc_blocks_x3<384>((hipStream_t)stream, x, pack, (size_t)master_elems * sizeof(float), master,
                                          reinterpret_cast<const EncBlockParams*>(table_ws), depth, 1e-6f, M, scratch, et)));
    return 0;
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
