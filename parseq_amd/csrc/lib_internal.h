// lib_internal.h — what the translation units of libparseq_hip.so share: error reporting, the device guard, the per-family
// event profiler, the model / plan objects behind include/parseq_hip.h and the launch helpers of the generic kernels.
// The library is split into lib_model.hip (model + plan objects), lib_infer.hip (encode / decode / forward),
// lib_train.hip (training step) and lib_ops.hip (per-kernel test entry points, resize, post-process) so that build() compiles
// them in parallel; kernels are templates or file-local (`static __global__`), so every unit carries the instantiations it launches.
#pragma once
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
#include "encoder_blocks_x3w.h"
#include "gemm.h"
#include "rowops.h"

using namespace pq;

// -------------------------------------------------------------------------------------------------------------------
// errors
// -------------------------------------------------------------------------------------------------------------------
inline thread_local char g_err[512] = "";
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
inline thread_local bool g_split = false;
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
    // parseq_train_encoder_forward: which storage mode (bit 0 bf16 shadows, bit 1 bf16-only slots) the record in `enc_record_ws` was written in
    int enc_record_mode = 0;
    const void* enc_record_ws = nullptr;
    // model constants of the training forward, built on first use: the one-launch encoder's block table (encoder_blocks.h; weight offsets relative
    // to the workspace's bf16 shadows, parameter offsets into the master) and the table of the one-launch weight-shadow kernel (train_ops.h)
    EncBlockParams* train_blocks_dev = nullptr;
    void* shadow_tab_dev = nullptr; int shadow_tiles = 0;
    // parseq_train_encoder_backward: one event per gradient segment (parseq_train_grad_segment), recorded on its stream as soon as that
    // part of the flat gradient buffer is final — the hook a data-parallel caller overlaps its bucket all-reduces with
    std::vector<hipEvent_t> grad_events;
    // the events belong to ONE step: parseq_train_decoder (the step's first gradient writer) invalidates them, a parseq_train_encoder_backward that ran to its end validates
    // them again — a caller that asks in between (the backward failed half-way, or never ran) must not get the previous step's events, which would let its collectives start
    // on gradients that are still being written
    bool grad_events_valid = false;
    // parseq_train_encoder_backward's second stream (the weight-gradient products run beside the dX / LayerNorm / attention chain) and the events
    // that order the two; created on first use — ONE SET PER CALLER STREAM (round 6: a step may run as micro-batches on several streams at
    // once, parseq_amd/train.py; two backwards that shared a side stream and its events would order each other's kernels)
    struct TrainSide { hipStream_t key = nullptr; hipStream_t side = nullptr; hipEvent_t ev[8] = {}; };
    std::vector<TrainSide> train_sides;

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


// plan
// -------------------------------------------------------------------------------------------------------------------
static __global__ void cvt_f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
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
static __global__ void split_pack_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;       // 4 consecutive elements
    if (i >= n) return;
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    uint2 hi, lo;
    split4(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, hi, lo);
    unsigned char* d = dst + (i >> 5) * 128 + (i & 31) * 2;
    *reinterpret_cast<uint2*>(d) = hi;
    *reinterpret_cast<uint2*>(d + 64) = lo;
}

static __global__ void cloze_mask_kernel(unsigned char* __restrict__ mask, int n, int ld) {
    // model.py:117,157: causal triu(1) with triu(2) cleared -> query i may not see key i + 1 only
    const int i = blockIdx.x, j = threadIdx.x;
    if (i < n && j < ld) mask[i * ld + j] = (j == i + 1) ? 1 : 0;
}

// ViTSTR sequence assembly: x[b][0] = cls_token + pos_embed[0]; x[b][1 + t] = xp[b][t] (patch rows, pos_embed already added).
static __global__ void insert_cls_kernel(const float* __restrict__ xp, const float* __restrict__ cls, const float* __restrict__ pos0,
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
    parseq_release_fn arena_release = nullptr;      // parseq_plan_create_ex: the caller's allocator gave the arena and takes it back (nullptr: hipMalloc / hipFree)
    void* arena_user = nullptr;
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
    bool kv24 = false;             // what kmem / vmem hold right now: f32 rows, or (bf16x3 one-launch encoder with its tail) the 24-bit rows of decoder_attn.h
    size_t kv_plane_elems = 0;     // elements of one K (or V) plane at max_batch: the u8 plane sits this many 2-byte elements behind the u16 plane
    bool kv24_enabled = getenv("PARSEQ_NO_KV24") == nullptr;      // diagnostics: keep f32 K / V rows in the bf16x3 mode
    int num_cus = 256;             // compute units of the device (tail-round avoidance of the one- and two-workgroup-per-CU kernels)
    bool fused_step = getenv("PARSEQ_NO_FUSED_STEP") == nullptr;   // diagnostics: fall back to the per-op AR step
    float* qfold = nullptr;        // [wbar E | cq E | bq2 E | c0 npos] of the split mid kernel (decoder_step.h dec_qfold_kernel), per weight set
    bool qsplit = getenv("PARSEQ_NO_QSPLIT") == nullptr;         // diagnostics: the mid kernel's start half on one workgroup per row tile
    bool fused_attn = getenv("PARSEQ_NO_FUSED_ATTN") == nullptr;   // diagnostics: qkv panel GEMM + attention + proj GEMM instead of encoder_attn_fused.h
    bool mlp_resident = getenv("PARSEQ_MLP_RELOAD") == nullptr;    // diagnostics: the fused MLP's first form (x re-read by the epilogue)
    bool fused_blocks = getenv("PARSEQ_NO_FUSED_BLOCKS") == nullptr;   // diagnostics: one launch per branch instead of encoder_blocks.h
    bool fused_x3 = getenv("PARSEQ_NO_FUSED_X3") == nullptr;
         // diagnostics: bf16x3 encoder through the per-op kernels instead of encoder_blocks_x3.h
    // The small-batch route (round 6): the one-launch bf16x3 encoder is one workgroup per image — a batch of B images occupies B of the 256
    // compute units for a whole twelve-block walk (3.3 ms) however small B is.  Up to this batch size the encoder runs as per-operation
    // launches instead, whose tiles spread one image's rows and columns over the device (profiles/r06_small_batch_route.md: batch 1
    // 4.10 -> 1.90 ms AR + 1, 3.32 -> 1.08 ms NAR + 3; crossover between 64 and 128).  Same arithmetic (three bf16 products per
    // product), another summation order: logits agree to 1e-5.  PARSEQ_SMALL_BATCH=<n> moves the threshold, 0 turns the route off.
    int small_batch_max = [] { const char* e = getenv("PARSEQ_SMALL_BATCH"); const int v = e ? atoi(e) : 64; return v < 0 ? 0 : v; }();
    bool x3_four_waves = getenv("PARSEQ_X3_FOUR_WAVES") != nullptr;
         // diagnostics: the bf16x3 one-launch encoder on four waves of 32 rows (encoder_blocks_x3.h) instead of eight of 16 (encoder_blocks_x3w.h); bit-identical results
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
        if constexpr (sizeof(T) == 4) {
            // bf16x3: the same kernel in the split arithmetic (three MFMAs per product); the fp32 mode keeps the scalar kernel below
            if (g_split && !getenv("PARSEQ_ATTN_GENERIC")) {
                const int nt32 = (tokens + 31) / 32;
#define PQ_ATTN_SN(NT)                                                                                                                  \
                if (nt32 == NT) {                                                                                                       \
                    static LdsAttr attr_;                                                                                               \
                    HIPCHK(attr_.ensure(reinterpret_cast<const void*>(attn_split_n_kernel<NT>), attn_split_n_lds<NT>()));               \
                    hipLaunchKernelGGL((attn_split_n_kernel<NT>), dim3(bh), dim3(64 * NT), attn_split_n_lds<NT>(), s, q, k, vt, ao, heads, tokens, scale); \
                    HIPCHK(hipGetLastError());                                                                                          \
                    return 0;                                                                                                           \
                }
                PQ_ATTN_SN(1) PQ_ATTN_SN(2) PQ_ATTN_SN(3) PQ_ATTN_SN(4) PQ_ATTN_SN(5) PQ_ATTN_SN(6) PQ_ATTN_SN(7)
#undef PQ_ATTN_SN
            }
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

// lib_encode.hip
int check_call(parseq_plan* p, int batch, int images_dtype);
int encode_dispatch(parseq_plan* p, const void* images, int images_dtype, int batch, float* memory_out, hipStream_t s);
