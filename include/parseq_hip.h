/*
 * parseq_hip.h — C ABI of libparseq_hip.so, the MI355X (gfx950) implementation of the PARSeq inference hot path.
 *
 * The reference (baudm/parseq @ /root/reference, strhub 1.2.0) is pure Python and has NO plugin / operator / FFI
 * interface for this path: the boundary it exposes is `PARSeq.forward(tokenizer, images, max_length)`
 * (strhub/models/parseq/model.py:105-169, called through strhub/models/parseq/system.py:87-88).  This header is the
 * interface a native backend sits behind; each entry point names the reference code it replaces.  The Python
 * binding a maintainer would add is shown in INTEGRATION.md; this repo's own binding is parseq_amd/_native.py
 * (ctypes).
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes.  No torch / ATen types.
 *   - every `const void*` / `float*` below that is documented "device" must be a pointer into HIP device memory
 *     of the device that was current when the model was created.  The caller owns all such buffers.
 *   - every call enqueues on the `hipStream_t` passed as `void* stream` (NULL = the default stream) and returns
 *     without synchronising, EXCEPT where stated (parseq_forward with PARSEQ_FLAG_TESTING and refine_iters == 0,
 *     which has to read one int back — the reference has a device->host sync per AR step at model.py:144).
 *   - return value: 0 on success, a negative PARSEQ_E_* code otherwise; parseq_last_error() gives the message
 *     (thread-local).  No exceptions cross the boundary.
 *   - a model and its plans are not internally locked: use one plan per host thread / stream.
 *   - devices: a model lives on the device that was current at parseq_model_create; every entry point that takes a model or
 *     a plan makes that device current for the duration of the call and restores the caller's (so `stream` must be a stream
 *     of the model's device).  The raw-pointer entry points (parseq_op_*, parseq_postprocess, parseq_resize_bicubic,
 *     parseq_cross_entropy, parseq_grad_norm) launch on the CURRENT device.  Per-kernel launch attributes are tracked per
 *     device, so one process may drive several GPUs / host threads (one plan each).
 *   - memory: the caller owns every tensor it passes; each plan owns ONE device arena (packed weights, decoder tables and all
 *     intermediates, about 0.9 GB at max_batch 512 in bf16) taken from hipMalloc (parseq_plan_create) or from the caller's
 *     allocator (parseq_plan_create_ex, ABI 7: "all workspace through the host framework's caching allocator", SURVEY.md
 *     section 8b), and the model one hipMalloc'ed buffer of fp32 master weights.  Nothing is allocated on the hot path; the
 *     arena's lifetime is the plan's.
 */
#ifndef PARSEQ_HIP_H_
#define PARSEQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARSEQ_ABI_VERSION 8

typedef struct parseq_model parseq_model;   /* weights of one PARSeq instance on one device */
typedef struct parseq_plan parseq_plan;     /* workspace + derived tables for (model, max_batch, precision) */

/* Constructor arguments of the reference model that shape the arithmetic
 * (strhub/models/parseq/model.py:33-49; values from configs/model/parseq.yaml, configs/main.yaml:9-10). */
typedef struct parseq_config {
    int32_t img_h, img_w;          /* 32, 128 */
    int32_t patch_h, patch_w;      /* 4, 8 */
    int32_t embed_dim;             /* 384 (PARSeq-S), 192 (PARSeq-Ti) */
    int32_t enc_depth, enc_heads, enc_mlp_ratio;   /* 12, 6 | 3, 4 */
    int32_t dec_depth, dec_heads, dec_mlp_ratio;   /* 1 (only value supported), 12 | 6, 4 */
    int32_t num_tokens;            /* len(tokenizer) = 97 for the 94-char set; the head predicts num_tokens - 2 classes */
    int32_t max_label_length;      /* 25 */
    int32_t bos_id, eos_id, pad_id;/* 95, 0, 96  (strhub/data/utils.py:107-111) */
    float enc_ln_eps, dec_ln_eps;  /* 1e-6 (timm ViT), 1e-5 (nn.LayerNorm default) */
    int32_t arch;                  /* PARSEQ_ARCH_PARSEQ (0) or PARSEQ_ARCH_VITSTR (1): the ViT encoder
                                    * with a class token and a per-token head (strhub/models/vitstr/model.py:20-28), no decoder;
                                    * the dec_* fields are ignored and the parameter keys are timm's (cls_token, pos_embed,
                                    * patch_embed.proj.*, blocks.N.*, norm.*, head.*) */
} parseq_config;

enum { PARSEQ_ARCH_PARSEQ = 0, PARSEQ_ARCH_VITSTR = 1 };

enum {
    PARSEQ_F32 = 0,    /* exact mode: f32 storage, v_mfma_f32_16x16x4_f32; parity target |dlogit| <= 1e-3 vs CPU fp32 */
    PARSEQ_BF16 = 1,   /* throughput mode: bf16 GEMM/attention operands, fp32 accumulate / residual / LayerNorm / softmax */
    PARSEQ_BF16X3 = 3, /* precision only (plans, parseq_op_linear / parseq_op_encoder_attention): exact-tolerance mode at matrix-core
                        * speed.  Storage is f32 exactly as in PARSEQ_F32 (images are passed as PARSEQ_F32 / _BF16 / _U8), but every
                        * GEMM / attention operand is split into a bf16 pair hi = bf16(v), lo = bf16(v - hi) and every product is
                        * evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_*_bf16 with fp32 accumulation: ~2^-17 relative error per
                        * product (bf16: 2^-9), 3 bf16 MFMAs per product instead of the f32 MFMA's 16x cost.  Meets the north
                        * star's |dlogit| <= 1e-3 with argmax-identical decodes (tests/test_hip_parity.py) */
    PARSEQ_U8 = 2      /* images_dtype only: raw 0..255 pixels, [batch, 3, H, W]; the patch-embed
                        * operand loader applies the reference transform's ToTensor + Normalize(0.5, 0.5)
                        * (strhub/data/module.py:78-81): ((v / 255) - 0.5) / 0.5 in f32, bit-identical to feeding the
                        * normalised f32 image (then rounded to bf16 in bf16 mode) */
};

enum {
    PARSEQ_FLAG_DECODE_AR = 1,   /* model.decode_ar (model.py:53,119): autoregressive vs one-shot NAR decoding */
    PARSEQ_FLAG_TESTING = 2,     /* max_length was None (model.py:106): batch-level early exit applies (model.py:144-145) */
    PARSEQ_FLAG_LATENCY = 4      /* (ABI 6) a hint, never a change of semantics: this forward has the device to itself (the reference's one call
                                  * at a time), so the AR step may spread its out_proj -> norm1 -> q-projection chain over three workgroups per
                                  * row tile (decoder_step.h DS_QS: a shorter dependent chain for ~2x the compute-unit time of that kernel).
                                  * Leave it clear when several forwards are in flight on different streams — there the compute units
                                  * the wider step takes are another batch's.  Results of the two forms differ by rounding only. */
};

enum {
    PARSEQ_E_INVALID = -1,       /* bad argument / unsupported configuration */
    PARSEQ_E_HIP = -2,           /* a HIP runtime call failed */
    PARSEQ_E_STATE = -3,         /* weights missing / not finalised */
    PARSEQ_E_ARCH = -4           /* device is not gfx950 */
};

int parseq_abi_version(void);
const char* parseq_last_error(void);

/* ---- model: replaces the nn.Module parameter storage of strhub/models/parseq/model.py:56-67 ---------------------- */

/* Validates the configuration (dec_depth == 1, head dims 64 / 32, embed_dim in {192, 384, 768}, 128 tokens) and the
 * device architecture.  Allocates device storage for the fp32 master weights. */
int parseq_model_create(const parseq_config* cfg, parseq_model** out);
void parseq_model_destroy(parseq_model* m);

/* Copies one parameter (fp32, contiguous, `numel` elements) from device memory.  `key` is the reference
 * state_dict key (e.g. "encoder.blocks.3.attn.qkv.weight", "decoder.layers.0.cross_attn.in_proj_weight",
 * "pos_queries"; full list: SURVEY.md section 8b), so a released checkpoint can be streamed in key by key.
 * Unknown key or wrong numel -> PARSEQ_E_INVALID. */
int parseq_model_set_param(parseq_model* m, const char* key, const float* device_ptr, int64_t numel, void* stream);

/* Number of parameters / i-th key and element count, for binding-side validation. */
int parseq_model_num_params(const parseq_model* m);
int parseq_model_param_info(const parseq_model* m, int index, const char** key, int64_t* numel);

/* ---- plan: per-(max_batch, precision) workspace, packed weights and batch-independent decoder tables ------------ */

/* All device memory the hot path needs is allocated here, never inside parseq_forward / parseq_encode.
 * Requires every parameter to have been set.  precision: PARSEQ_F32, PARSEQ_BF16 or PARSEQ_BF16X3.  (Re)packs weights for `precision`; call parseq_plan_refresh after
 * parameters change. */
int parseq_plan_create(parseq_model* m, int max_batch, int precision, void* stream, parseq_plan** out);
/* (ABI 7) The same with the plan's ONE device arena taken from the caller's allocator instead of hipMalloc — "all workspace through the
 * host framework's caching allocator" (what the reference gets for free from torch: every intermediate of
 * strhub/models/parseq/model.py:86-169 is a caching-allocator block).  alloc(bytes, user) is called exactly once, inside this call, on the
 * model's device, with bytes == what parseq_plan_workspace_bytes will report; it returns device memory aligned to 256 bytes or NULL
 * (-> PARSEQ_E_HIP).  release(ptr, user) is called exactly once from parseq_plan_destroy (or from this call if a later step of the
 * creation fails); the caller's release must not recycle the block while work of this plan is still in flight on any stream (hipFree
 * synchronises implicitly; a caching allocator does not).  alloc == NULL and release == NULL: hipMalloc / hipFree (parseq_plan_create).
 * A torch binding hands c10::hip::HIPCachingAllocator::raw_alloc / raw_delete; the ctypes binding of this repository allocates a uint8
 * torch tensor per plan (parseq_amd/_native.py torch_plan_allocator). */
typedef void* (*parseq_alloc_fn)(size_t bytes, void* user);
typedef void (*parseq_release_fn)(void* ptr, void* user);
int parseq_plan_create_ex(parseq_model* m, int max_batch, int precision, void* stream, parseq_alloc_fn alloc, parseq_release_fn release,
                          void* user, parseq_plan** out);
int parseq_plan_refresh(parseq_plan* p, void* stream);
void parseq_plan_destroy(parseq_plan* p);
size_t parseq_plan_workspace_bytes(const parseq_plan* p);

/* Optional per-kernel-family timing: while enabled, every launch is bracketed by HIP events on the caller's stream
 * (this perturbs throughput: use a separate pass).  parseq_plan_get_profile(index) returns 0 and fills the family name,
 * accumulated milliseconds and launch count since profiling was (re-)enabled, or returns 1 when index is past the last
 * family; it synchronises on the recorded events. */
int parseq_plan_set_profiling(parseq_plan* p, int enable);
int parseq_plan_get_profile(parseq_plan* p, int index, const char** name, double* total_ms, int64_t* launches);

/* ---- hot path ---------------------------------------------------------------------------------------------------- */

/* model.PARSeq.encode (model.py:83-84 -> modules.py:163-165 -> timm ViT.forward_features).
 * images: device, [batch, 3, img_h, img_w], contiguous; normalised fp32 (images_dtype = PARSEQ_F32) or bf16 (PARSEQ_BF16),
 * or raw uint8 pixels (PARSEQ_U8, normalised on the fly).
 * memory_out: device fp32 [batch, tokens, embed_dim], or NULL to keep the result only inside the plan. */
int parseq_encode(parseq_plan* p, const void* images, int images_dtype, int batch, float* memory_out, void* stream);

/* model.PARSeq.forward (model.py:105-169): encode, AR loop or NAR pass, `refine_iters` cloze refinements.
 * num_steps = min(max_length, max_label_length) + 1 (model.py:107-110), i.e. 26 by default.
 * logits_out: device fp32 [batch, num_steps, num_tokens - 2], contiguous.  Rows [*, 0:L, *] are valid, where L is
 * written to *out_len (host int): L = num_steps, except AR + PARSEQ_FLAG_TESTING + refine_iters == 0, where L is the
 * early-exit length of model.py:144-145 and the call synchronises the stream to read it. */
int parseq_forward(parseq_plan* p, const void* images, int images_dtype, int batch, int flags, int refine_iters,
                   int num_steps, float* logits_out, int* out_len, void* stream);

/* model.PARSeq.decode + head for the contexts the reference's forward() builds (model.py:86-103, 138, 152, 167),
 * exposed for per-stage parity tests: `memory` must have been produced by parseq_encode on this plan (its K/V
 * projection is cached in the plan).  tokens: device int32 [batch, ctx_len] (tgt_in).  Queries are
 * pos_queries[q_start : q_start + q_len].  query_mask: device uint8 [num_steps? no: (max_label_length + 1)] x ctx_len
 * rows indexed by absolute query position, or NULL; key_padding_mask: device uint8 [batch, ctx_len] or NULL
 * (non-zero = masked, torch semantics).  logits_out: device fp32 [batch, q_len, num_tokens - 2]. */
int parseq_decode_logits(parseq_plan* p, const int32_t* tokens, int batch, int ctx_len, int q_start, int q_len,
                         const uint8_t* query_mask, const uint8_t* key_padding_mask, float* logits_out, void* stream);

/* model.PARSeq.decode (model.py:86-103): the same pass as parseq_decode_logits, additionally returning what `decode`
 * returns in the reference — the decoder output after decoder.norm (modules.py:124), fp32 [batch, q_len, embed_dim] — so that
 * `model.head(model.decode(...))` keeps working.  logits_out is written too (head of the same pass, library arithmetic). */
int parseq_decode_hidden(parseq_plan* p, const int32_t* tokens, int batch, int ctx_len, int q_start, int q_len,
                         const uint8_t* query_mask, const uint8_t* key_padding_mask, float* hidden_out, float* logits_out,
                         void* stream);

/* model.PARSeq.decode with a caller-supplied query stream (model.py:100-102: `tgt_query` need not be a slice of pos_queries):
 * query: device fp32 [batch, q_len, embed_dim].  norm_q + the query projection run at call time, self-attention scores are
 * computed against the content-key table, and the residual stream starts from `query` itself (modules.py:60-66).
 * query_mask: device uint8 [q_len, ctx_len] or NULL.  hidden_out: fp32 [batch, q_len, embed_dim] or NULL; logits_out as above. */
int parseq_decode_query(parseq_plan* p, const int32_t* tokens, int batch, int ctx_len, const float* query, int q_len,
                        const uint8_t* query_mask, const uint8_t* key_padding_mask, float* hidden_out, float* logits_out,
                        void* stream);

/* model.PARSeq.decode's `memory` argument (model.py:89): projects a caller-supplied encoder output — device fp32
 * [batch, tokens, embed_dim], as parseq_encode returns it (after the encoder's final norm) — to the decoder's cross-attention
 * K / V, replacing the ones cached by the last parseq_encode / parseq_forward on this plan.  The decode entry points then
 * attend to THIS memory until the next encode / forward / set_memory. */
int parseq_set_memory(parseq_plan* p, const float* memory, int batch, void* stream);

/* ViTSTR.forward (strhub/models/vitstr/system.py:76-82 -> vitstr/model.py:20-28) for a model created with
 * arch = PARSEQ_ARCH_VITSTR: encoder with class token, head on tokens [1, num_steps], i.e. logits_out fp32
 * [batch, num_steps, num_tokens - 2] with num_steps = min(max_length, max_label_length) + 1.  images as parseq_forward. */
int parseq_vitstr_forward(parseq_plan* p, const void* images, int images_dtype, int batch, int num_steps, float* logits_out,
                          void* stream);

/* ---- input resize ------------------------------------------------------------------- */

/* One RGB image, HWC uint8: `data` is a DEVICE pointer, row_stride in bytes (>= 3 * width). */
typedef struct {
    const uint8_t* data;
    int32_t height, width;
    int64_t row_stride;
} parseq_image_desc;

/* The resize step of the reference's input transform, T.Resize(img_size, BICUBIC) on a PIL image
 * (strhub/data/module.py:77, read.py:41-43): Pillow's 8-bit ImagingResample, bit-exact (double-precision weights, 22-bit
 * fixed point, horizontal pass stored as uint8 before the vertical pass).  images: HOST array of `batch` descriptors
 * (image sizes may differ); out: device uint8 [batch, 3, out_h, out_w] — exactly the PARSEQ_U8 input of parseq_forward /
 * parseq_encode, which applies ToTensor + Normalize(0.5, 0.5) in its patch-embed loader; workspace: device scratch of
 * parseq_resize_workspace_bytes(batch) bytes (the descriptors are copied there on `stream`). */
size_t parseq_resize_workspace_bytes(int batch);
int parseq_resize_bicubic(const parseq_image_desc* images, int batch, int out_h, int out_w, uint8_t* out, void* workspace, void* stream);

/* ---- post-processing --------------------------------------------------------------- */

/* Numeric half of `preds, probs = tokenizer.decode(logits.softmax(-1))` (strhub/models/base.py:132-137,
 * strhub/data/utils.py:79-99 greedy max per position, :120-129 truncation at the first EOS) on the device, so a caller
 * moves B*L ids + B lengths (+ probabilities) to the host instead of B*L*C probabilities and one .tolist() per row.
 * logits: device fp32 [batch, L, C] contiguous (L <= 64).  Outputs (device):
 *   ids_out      int32 [batch, L]  arg-max class per position (first maximum), for every position
 *   lengths_out  int32 [batch]     index of the first position whose id is eos_id, or L: the label is ids_out[b, :len]
 *   probs_out    fp32  [batch, L]  max soft-max probability per position (NULL to skip); the reference's per-label
 *                                  probability tensor is probs_out[b, :min(len + 1, L)] (it keeps the EOS probability)
 *   confidence_out fp32 [batch]    product of that tensor (base.py:137), NULL to skip */
int parseq_postprocess(const float* logits, int batch, int L, int C, int eos_id, int32_t* ids_out, int32_t* lengths_out,
                       float* probs_out, float* confidence_out, void* stream);

/* Validation loss of strhub/models/base.py:194-201 (CrossEntropySystem.forward_logits_loss):
 * F.cross_entropy(logits.flatten(end_dim=1), targets.flatten(), ignore_index) with mean reduction, and the number of
 * non-ignored targets.  logits: device fp32 [rows, C]; targets: device int32 [rows] (class index or ignore_index);
 * loss_out / numel_out: device scalars; workspace: device, `rows` floats.  Deterministic (fixed summation order). */
int parseq_cross_entropy(const float* logits, const int32_t* targets, int rows, int C, int ignore_index, float* loss_out,
                         int32_t* numel_out, float* workspace, void* stream);

/* ---- "next" row N3: training step, decoder side (strhub/models/parseq/system.py:168-199 + loss.backward()) ---------- */

/* Offset (in floats) of parameter `index` inside a gradient buffer, and the buffer's length: gradients are laid out like
 * the model's fp32 master weights, every tensor starting at a multiple of 8 floats.  -1 for a bad index. */
int64_t parseq_model_param_offset(const parseq_model* m, int index);
int64_t parseq_model_grad_elems(const parseq_model* m);
/* Arithmetic of the training step's matrix products (reference: `precision: bf16-mixed`, /root/reference/train.py:62-64):
 * PARSEQ_F32 (default) = exact fp32 products on v_mfma_f32_16x16x4_f32; PARSEQ_BF16 = both operands of every aligned Linear
 * product (forward, dX, dW) rounded to bfloat16 on their way into LDS, fp32 accumulate, fp32 master weights / activations /
 * gradients in memory.  Attention products, LayerNorm, soft-max, loss and the optimiser stay fp32 in both modes. */
int parseq_model_set_train_precision(parseq_model* m, int precision);

/* Loss of the K-permutation training objective (system.py:168-199, dropout off) for a batch whose encoder output is
 * `memory`, and its gradients — what `loss.backward()` leaves in `.grad` of every decoder-side parameter (decoder.*,
 * head.*, text_embed.*, pos_queries; system.py:183-196 through model.py:86-103, modules.py:55-125) and the gradient
 * w.r.t. `memory` that the encoder backward starts from.  fp32 throughout, computed from the model's master weights;
 * deterministic.
 *   memory            device fp32 [batch, tokens, embed_dim]             (parseq_encode output)
 *   tokens            device int32 [batch, ctx_len]   tgt[:, :-1] of tokenizer.encode(labels)            (system.py:176)
 *   targets           device int32 [2][batch * ctx_len]: tgt[:, 1:] flattened, and the same with <eos> replaced by <pad>
 *                     (used from the third permutation on, system.py:191-195)
 *   key_padding_mask  device uint8 [batch, ctx_len]   (tgt_in == pad) | (tgt_in == eos)                   (system.py:179)
 *   query_masks       device uint8 [num_perms][ctx_len][ctx_len]   generate_attn_masks(perm)[1]           (system.py:152-166)
 *   total_targets     sum over the permutations of their count of non-<pad> targets (the loss denominator, :189,196)
 *   dropout_p, seed   dropout probability of the decoder (configs/model/parseq.yaml:21; 0 = off, the evaluation-mode step the
 *                     parity tests use) and the 64-bit seed of this step's masks.  Eight sites per permutation pass, as in
 *                     the reference (model.py:99-102 embeddings and queries, dropped afresh in every pass; modules.py:33-43,
 *                     70-79 both attentions' probabilities, both projections, the MLP's hidden layer and output).  Masks
 *                     come from a counter-based generator (train_ops.h:drop_factor), not torch's Philox stream: with the
 *                     same masks the gradients are exact (tests), against the reference's run they agree in distribution.
 *   loss_out          device fp32 [1 + num_perms]: the loss, then each permutation's mean cross-entropy
 *   grads             device fp32 [parseq_model_grad_elems]: ACCUMULATED into (zero it for a fresh step); encoder slots untouched
 *   dmemory           device fp32 [batch, tokens, embed_dim]: written
 *   workspace         device, parseq_train_decoder_workspace_bytes(...) bytes; after the call it holds the intermediates of
 *                     the last permutation (parseq_train_decoder_workspace_offset names them; used by the parity tests).
 * The num_perms passes share every weight and differ in masks, dropout sites and (after two passes) targets only, so they run as
 * ONE batch of num_perms * batch images (round 3): same masks bit for bit, same per-pass losses, gradients equal to the
 * one-pass-after-the-other form up to fp32 summation order; the workspace holds num_perms copies of the per-pass buffers (4.5 GB
 * at batch 384, 6 passes).  The environment variable PARSEQ_TRAIN_PERM_GROUP=g (read by both functions below) runs them g at a time. */
size_t parseq_train_decoder_workspace_bytes(const parseq_model* m, int batch, int ctx_len, int num_perms);
int64_t parseq_train_decoder_workspace_offset(const parseq_model* m, int batch, int ctx_len, int num_perms, const char* name);
int parseq_train_decoder(parseq_model* m, const float* memory, const int32_t* tokens, const int32_t* targets,
                         const uint8_t* key_padding_mask, const uint8_t* query_masks, int batch, int ctx_len, int num_perms,
                         int total_targets, float dropout_p, uint64_t seed, float* loss_out, float* grads, float* dmemory,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Encoder half of the training step (timm VisionTransformer blocks, strhub/models/parseq/modules.py:128-165): a forward in
 * fp32 from the master weights that keeps in `workspace` what the backward needs (per block: the residual stream before
 * it, qkv, the attention output, the stream after the attention residual, the fc1 pre-activation), and the backward from
 * `dmemory` (parseq_train_decoder) to the gradient of every encoder.* parameter, ACCUMULATED into `grads`.
 * images: device fp32 [batch, 3, H, W], normalised as the reference's transform leaves them.
 * Streams: in the bf16-operand mode parseq_train_encoder_backward runs the blocks' weight-gradient products on a second, library-owned
 * non-blocking stream beside the chain on `stream` (PARSEQ_TRAIN_ONE_STREAM=1 turns that off) — one such stream per distinct caller stream
 * (up to eight per model), so independent training chains may share a model on different streams (tests/test_training.py
 * test_micro_batched_step_equals_the_one_piece_step).  Every block ends with `stream` waiting
 * for it, so on return everything is ordered behind `stream` as usual — on an error return as well: a failure inside a block
 * synchronises the second stream before it is reported, so `grads` and `workspace` are not in use behind the caller's back. */
size_t parseq_train_encoder_workspace_bytes(const parseq_model* m, int batch);
int parseq_train_encoder_forward(parseq_model* m, const float* images, int batch, float* memory_out, void* workspace,
                                 size_t workspace_bytes, void* stream);
int parseq_train_encoder_backward(parseq_model* m, const float* dmemory, int batch, float* grads, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* Gradient segments (ABI 5): the hook for overlapping the data-parallel all-reduce with the encoder's backward — what DDP's bucketed
 * reducer does under the reference's Trainer(strategy=DDPStrategy(...), reference train.py:65-71, 88-96).  The flat gradient buffer
 * becomes final piecewise: the decoder's part before parseq_train_encoder_backward starts, then the encoder's blocks from the last to
 * the first.  parseq_train_grad_segments = number of segments (enc_depth + 1; 0 for ViTSTR); segment `index` (in completion order) is
 * the element range [*begin, *end) of the buffer — the ranges tile [0, parseq_model_grad_elems) — and *event (may be NULL) is a
 * hipEvent_t the most recent parseq_train_encoder_backward recorded on its stream right after the last kernel that writes the range.
 * parseq_stream_wait_event makes `stream` wait for it (hipStreamWaitEvent), so that a collective enqueued on that stream starts as
 * soon as its bucket is final while the backward keeps running.  The events belong to ONE step (ABI 7): parseq_train_decoder — the
 * step's first gradient writer — invalidates them, a parseq_train_encoder_backward that returns 0 validates them; asking for an
 * event in between (the backward failed, or was never called for this step) returns PARSEQ_E_STATE instead of the previous step's
 * already-signalled event.  The ranges (event == NULL) are always available. */
int parseq_train_grad_segments(const parseq_model* m);
int parseq_train_grad_segment(parseq_model* m, int index, int64_t* begin, int64_t* end, void** event);
int parseq_stream_wait_event(void* stream, void* event);

/* Data-parallel sharding (ABI 8; SURVEY.md §8(e): the path shards over independent crops, reference bench.py / test.py loop over batches
 * per device under Lightning).  The contiguous split the Python mirror uses (parseq_amd/parallel.py shard_bounds): `world` shards of n
 * items whose sizes differ by at most one, shard `rank` = [*begin, *end).  A caller without torch shards its batch with this, runs
 * parseq_forward on its own shard with its own plan, and exchanges the [b_local, L, C] logits itself — ONE all-gather on its own RCCL
 * communicator when n is a multiple of world (INTEGRATION.md shows the calls); there is no collective inside the library.
 * Returns PARSEQ_E_INVALID for n < 0, world < 1 or rank outside [0, world). */
int parseq_shard_bounds(int64_t n, int world, int rank, int64_t* begin, int64_t* end);

/* Optimiser half of the training step (strhub/models/base.py:98-107: timm create_optimizer_v2('adamw') = torch.optim.AdamW;
 * configs/main.yaml:39 gradient_clip_val = torch.nn.utils.clip_grad_norm_), over the flat buffers:
 *   parseq_grad_norm   norm_out[0] = L2 norm of grads[0..n)  (workspace: 1024 floats); deterministic
 *   parseq_adamw_step  one AdamW update of the model's fp32 master weights IN PLACE from `grads`; exp_avg / exp_avg_sq are the
 *                      caller-owned moment buffers [parseq_model_grad_elems] (zero before step 1); `step` counts from 1;
 *                      decay_flags: host int32 [num_params], non-zero = weight decay applies to that tensor (NULL = none);
 *                      grad_norm: device scalar from parseq_grad_norm or NULL — when given the gradient is scaled by
 *                      min(1, max_norm / (norm + 1e-6)) on the fly (no host round trip).  Plans built on the model must be
 *                      refreshed (parseq_plan_refresh) before the next inference call.
 *   parseq_model_get_param  copies one parameter of the master weights out (device fp32), the inverse of parseq_model_set_param
 *   parseq_model_get_params copies EVERY parameter out in one launch: device_ptrs is a HOST array of `count` = parseq_model_num_params
 *                      device pointers in parseq_model_param_info order, each to that parameter's numel floats (what the reference's
 *                      optimizer.step() leaves in the module's tensors, strhub/models/base.py:98-107) */
int parseq_grad_norm(const float* grads, int64_t n, float* norm_out, float* workspace, void* stream);
int parseq_adamw_step(parseq_model* m, const float* grads, float* exp_avg, float* exp_avg_sq, const int32_t* decay_flags, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_norm, float max_norm,
                      void* stream);
int parseq_model_get_param(const parseq_model* m, const char* key, float* device_ptr, int64_t numel, void* stream);
int parseq_model_get_params(parseq_model* m, float* const* device_ptrs, int count, void* stream);

/* ---- single operators, exported so each kernel is parity-tested through the C ABI ------------------------------- */

/* y = LayerNorm(x) over the last dim `E` (192 | 384 | 768); x fp32 [rows, E]; y in out_dtype. */
int parseq_op_layernorm(const float* x, const float* w, const float* b, void* y, int out_dtype, int rows, int E,
                        float eps, void* stream);
/* C = A W^T + bias.  A [M, K] and W [N, K] in `dtype`, bias fp32 [N] or NULL, C fp32 [M, N] (act = 0) or
 * C in `dtype` with exact-erf GELU applied (act = 1).  K must be a multiple of 8. */
int parseq_op_linear(const void* A, const void* W, const float* bias, void* C, int dtype, int act, int M, int N, int K,
                     void* stream);
/* C[M, N] (fp32) = LayerNorm(x[M, 384]; gamma, beta, eps) W^T + bias with the LayerNorm fused into the A-operand loader of the tile
 * GEMM (the decoder's q-projection / linear1 / head form).  dtype PARSEQ_F32 (W fp32 [N, 384]) or PARSEQ_BF16X3 (W from
 * parseq_op_split_pack). */
int parseq_op_ln_linear(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, float* C, int dtype,
                        int M, int N, float eps, void* stream);
/* bf16x3, both operands pre-split (the encoder's big-M form): ws[M * 384 * 4 bytes] receives LayerNorm(x[M, 384]; gamma, beta, eps) as
 * block-planar hi | lo bf16 pairs (layernorm_split_kernel), then C[M, N] (fp32) = ws W^T + bias with W from parseq_op_split_pack,
 * through the direct-to-LDS GEMM that reads both operands as pairs.  act != 0: C is instead written as gelu(...) in the same
 * block-planar pair layout (N a multiple of 32; C is N * 4 bytes per row) — the fc1 epilogue of that path. */
int parseq_op_ln_linear_pairs(const float* x, const float* gamma, const float* beta, const void* W, const float* bias, void* C, void* ws,
                              int act, int M, int N, float eps, void* stream);
/* dtype = PARSEQ_BF16X3: A is f32 [M, K], K a multiple of 32; W must be the block-planar hi / lo copy of the f32 weight that
 * parseq_op_split_pack(src f32 [numel], dst [numel * 4 bytes]) produces (numel a multiple of 32); C as for PARSEQ_F32. */
int parseq_op_split_pack(const float* src, void* dst, int64_t numel, void* stream);
/* Same as parseq_op_linear with an explicit tile configuration (tools/gemm_bench.py sweeps these; ids in lib_ops.hip). */
int parseq_op_linear_cfg(const void* A, const void* W, const float* bias, void* C, int dtype, int act, int M, int N, int K,
                         int cfg, void* stream);
/* out[M, N] (bf16) = gelu(LayerNorm(x[M, 384]; gamma, beta, eps 1e-6) W^T + bias) through the register-resident-A panel
 * kernel; W bf16 [N, 384], N a multiple of 128.  variant must be 0 (the ablation variants of rounds 1-2 were removed; the parameter stays for ABI 4). */
int parseq_op_ln_linear_gelu(const float* x, const float* gamma, const float* beta, const void* W, const float* bias,
                             void* out, int M, int N, int variant, void* stream);
/* In place x[M, 384] (fp32) += fc2(gelu(fc1(LayerNorm(x; gamma, beta, eps 1e-6)))) through the fused MLP kernel:
 * W1 bf16 [1536, 384], b1 fp32 [1536], W2 bf16 [384, 1536], b2 fp32 [384]  (timm Block: x + mlp(norm2(x))). */
int parseq_op_mlp(float* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                  const float* b2, int M, void* stream);
/* In place x[M, 384] (fp32) += proj(attention(qkv(LayerNorm(x; gamma, beta, eps 1e-6)))) through the fused attention-branch kernel
 * (timm Block: x + attn(norm1(x)), 6 heads of 64, one image = 128 consecutive rows per workgroup; M a multiple of 128):
 * Wqkv bf16 [1152, 384] (q | k | v rows, head-major), bqkv fp32 [1152], Wproj bf16 [384, 384], bproj fp32 [384].
 * variant 0 = encoder_attn_fused.h; 1 = the phase function the one-launch encoder is built from (encoder_blocks.h attn_branch_kernel). */
int parseq_op_attn_fused(float* x, const float* gamma, const float* beta, const void* Wqkv, const float* bqkv, const void* Wproj,
                         const float* bproj, int M, int variant, void* stream);
/* `depth` encoder blocks in one launch (encoder_blocks.h): in place on x[M, 384] (fp32), M a multiple of 128 (one image = 128
 * consecutive rows per workgroup).  block_ptrs: HOST array of depth * 12 DEVICE pointers, per block in this order: norm1 weight,
 * norm1 bias (fp32 [384]), Wqkv bf16 [1152, 384], bqkv fp32 [1152], Wproj bf16 [384, 384], bproj fp32 [384], norm2 weight, norm2
 * bias, W1 bf16 [1536, 384], b1 fp32 [1536], W2 bf16 [384, 1536], b2 fp32 [384].  table_ws: device scratch of depth * 48 bytes.
 * Test hook (the product path builds the table once per plan); uploads the table synchronously. */
int parseq_op_enc_blocks(float* x, const void* const* block_ptrs, int depth, int M, void* table_ws, void* stream);
/* The head and the tail of the one-launch encoder with no blocks in between (encoder_blocks.h patch_head / kv_phase; M a multiple of
 * 128 = whole 32 x 128 crops of 8 x 16 patches).  images != NULL ([M / 128, 3, 32, 128], images_dtype PARSEQ_F32 / PARSEQ_BF16 /
 * PARSEQ_U8): x = patches(images) Wpe^T + posb in the accumulators — timm PatchEmbed (4, 8) + bias + pos_embed, Wpe bf16 [384, 96] =
 * the Conv2d weight flattened (k = 32 c + 8 ky + kx), posb fp32 [128, 384] = pos_embed + bias; otherwise x[M, 384] (fp32) is loaded.
 * kmem != NULL: the launch ends with K | V = LayerNorm(x; norm_w, norm_b, eps 1e-6) Wkv^T + bkv (Wkv bf16 [768, 384], bkv fp32 [768];
 * strhub/models/parseq/modules.py:33-34 on timm's final norm) written as bf16 [M / 128][12][128][32] into kmem / vmem; otherwise x is
 * stored.  Test hook for the two phases the product path only runs inside parseq_forward. */
int parseq_op_enc_head_tail(float* x, const void* images, int images_dtype, const void* wpe, const float* posb, const float* norm_w,
                            const float* norm_b, const void* wkv, const float* bkv, void* kmem, void* vmem, int M, void* stream);
/* `depth` encoder blocks in one launch in the bf16x3 arithmetic (encoder_blocks_x3.h), in place on x[M, 384] (fp32), M a multiple of 128.
 * master: ONE fp32 device buffer holding every parameter of the blocks, each tensor on a 32-element boundary; pack: its block-planar
 * hi | lo copy (parseq_op_split_pack over the whole buffer, master_elems * 4 bytes); offsets: HOST array of depth * 12 element offsets
 * into master, per block: norm1 weight, norm1 bias, Wqkv [1152, 384], bqkv, Wproj [384, 384], bproj, norm2 weight, norm2 bias,
 * W1 [1536, 384], b1, W2 [384, 1536], b2.  table_ws: device scratch of depth * 48 bytes; scratch: device, M / 128 * 393216 bytes.
 * kmem / vmem != NULL (fp32 [M / 128][12][128][32]) with tail_offsets (HOST: final norm weight, bias, Wkv [768, 384], bkv [768]):
 * x is not stored; the launch ends with the decoder's K | V = LayerNorm(x) Wkv^T + bkv (timm forward_features' norm, modules.py:33-34).
 * Test hook (the product path builds the tables once per plan); uploads the table synchronously. */
int parseq_op_enc_blocks_x3(float* x, const float* master, const void* pack, int64_t master_elems, const uint32_t* offsets, int depth,
                            int M, void* table_ws, float* scratch, const uint32_t* tail_offsets, float* kmem, float* vmem, void* stream);
/* The same launch through the kernel parseq_forward runs since ABI 7 (encoder_blocks_x3w.h: eight waves of 16 rows per workgroup, two per
 * SIMD, instead of four of 32); same arguments, bit-identical results.  parseq_op_enc_blocks_x3 stays on the four-wave kernel as the
 * reference of that identity (PARSEQ_X3_FOUR_WAVES=1 puts it back under parseq_forward). */
int parseq_op_enc_blocks_x3w(float* x, const float* master, const void* pack, int64_t master_elems, const uint32_t* offsets, int depth,
                             int M, void* table_ws, float* scratch, const uint32_t* tail_offsets, float* kmem, float* vmem, void* stream);
/* The forms of parseq_op_mlp: variant 0 = x re-read by the epilogue; 10 = x resident in the fc2 accumulators (the form the per-layer
 * encoder path uses); 11 = the phase function the one-launch encoder is built from (encoder_blocks.h mlp_branch_kernel). */
int parseq_op_mlp_variant(float* x, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2,
                          const float* b2, int M, int variant, void* stream);
/* Encoder attention for `bh` (image, head) pairs: q, k [bh, 128, 64], vt [bh, 64, 128] in `dtype`;
 * out [bh / heads * 128, heads * 64] in `dtype`.  dtype = PARSEQ_BF16X3: f32 tensors, split-bf16 products. */
int parseq_op_encoder_attention(const void* q, const void* k, const void* vt, void* out, int dtype, int bh, int heads,
                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARSEQ_HIP_H_ */
