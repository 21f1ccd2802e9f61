"""CPU oracle for the PARSeq inference hot path.   *** TEST INFRASTRUCTURE — NOT A PRODUCT PATH ***

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
`parseq_amd/` never does: the product path is the HIP library and fails loudly without it.

What this is: a plain-torch (fp32, CPU) restatement of the algorithm on the reference's hot path,
written as pure functions over a flat `state_dict` with the reference's key names.  Each function
cites the reference lines it restates (paths relative to the upstream repo root):

  encode()            strhub/models/parseq/model.py:83-84 -> modules.py:163-165 -> timm ViT.forward_features
                      (timm==0.9.16 is third-party and NOT vendored in the reference; restated from its
                      published semantics, see oracle/timm_standin.py and DESIGN.md section 3)
  token_embedding()   strhub/models/parseq/modules.py:168-176
  mha()               torch.nn.functional.multi_head_attention_forward, need_weights=True branch
                      (torch/nn/functional.py:6576-6606 in the torch 2.10 shipped in this image), as driven by
                      strhub/models/parseq/modules.py:33-34,70-75
  decoder_layer()     strhub/models/parseq/modules.py:55-98 (forward_stream + forward, update_content=False)
  decode()            strhub/models/parseq/model.py:86-103 + modules.py:101-125 (Decoder, final LayerNorm)
  forward()           strhub/models/parseq/model.py:105-169 (AR loop, NAR branch, cloze refinement, early exit)
  attn_masks_from_perm(), training_loss()
                      strhub/models/parseq/system.py:152-199 (row N3: permutation masks, K-permutation loss; pinned by
                      oracle/make_golden_train.py, which runs the reference's own system.py, loss and gradients)

Pinning: `oracle/make_golden.py` executes the reference's own model.py/modules.py (on the timm stand-in)
in the build container and freezes inputs + outputs under tests/golden/; tests/test_oracle.py requires this
oracle to reproduce them.  The decoder/AR/refinement logic is therefore pinned by execution of the
reference; the ViT encoder arithmetic is pinned only against the stand-in ("parity unpinned" at the timm
boundary) plus an independent cross-check against `transformers`' ViTLayer.

Two arithmetic modes:
  rounding=None    exact fp32 — the parity target for the library's fp32 mode (|dlogit| <= 1e-3, same argmax).
  rounding='bf16'  the same algorithm with every GEMM operand rounded to bfloat16 (fp32 accumulate) at exactly
                   the points where the HIP bf16 path stores bf16 — a rounding-aware oracle for the
                   throughput mode (see DESIGN.md section 6 for the list of rounding points).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor


@dataclass
class OracleConfig:
    """Constructor arguments of the reference model (strhub/models/parseq/model.py:33-49) that shape the math."""
    img_size: Sequence[int] = (32, 128)
    patch_size: Sequence[int] = (4, 8)
    embed_dim: int = 384
    enc_num_heads: int = 6
    enc_mlp_ratio: int = 4
    enc_depth: int = 12
    dec_num_heads: int = 12
    dec_mlp_ratio: int = 4
    dec_depth: int = 1
    num_tokens: int = 97            # len(Tokenizer(charset_train)): [E] + 94 chars + [B] + [P]
    max_label_length: int = 25
    eos_id: int = 0
    bos_id: int = 95
    pad_id: int = 96
    enc_ln_eps: float = 1e-6        # timm ViT: partial(nn.LayerNorm, eps=1e-6)
    dec_ln_eps: float = 1e-5        # modules.py:31 layer_norm_eps default; model.py:60 nn.LayerNorm default

    @property
    def num_patches(self) -> int:
        return (self.img_size[0] // self.patch_size[0]) * (self.img_size[1] // self.patch_size[1])


# ----------------------------------------------------------------------------------------------------------
# rounding helpers
# ----------------------------------------------------------------------------------------------------------

def _r(x: Tensor, rounding, site: str = '') -> Tensor:
    """Round to a storage type and come back to fp32 (identity in exact mode).  `rounding` is None, a type name applied at every
    rounding point ('bf16': the throughput mode), or a dict {site: type name} that rounds only the named sites (the cheaper-exact-mode
    study of tools/cheap_exact_study.py): 'enc.act' / 'enc.w' / 'dec.act' / 'dec.w' (the two operands of the Linear products),
    'img', 'enc.qkv', 'enc.p' (encoder attention operands), 'dec.kv' (the decoder's stored K / V; 'dec.k' / 'dec.v' one of them)."""
    if isinstance(rounding, dict):
        rounding = rounding.get(site)
    if rounding is None:
        return x
    if rounding == 'bf16':
        return x.to(torch.bfloat16).to(torch.float32)
    if rounding == 'fp16':
        return x.to(torch.float16).to(torch.float32)
    if rounding == 'bf16+8':      # bf16 hi + an 8-bit residual on the hi's exponent: 16 significant bits
        hi = x.to(torch.bfloat16).to(torch.float32)
        ulp = torch.exp2(torch.floor(torch.log2(hi.abs().clamp_min(1e-38))) - 7.0)      # bf16 ulp of hi
        return hi + torch.round((x - hi) / ulp * 256.0) * ulp / 256.0
    raise ValueError(rounding)


# product emulations registered by studies (tools/cheap_exact_study.py): name -> f(x, w) = the value a candidate matrix-core scheme computes
# for x W^T (e.g. integer-slice or scaled-fp8 products); selected per site with rounding = {'enc.prod': name} / {'dec.prod': name}
PRODUCT_EMULATIONS = {}


def _linear(x: Tensor, w: Tensor, b: Optional[Tensor], rounding, site: str = 'dec') -> Tensor:
    """y = x W^T + b with both GEMM operands rounded (fp32 accumulate), bias added in fp32."""
    if isinstance(rounding, dict) and rounding.get(site + '.prod') is not None:
        y = PRODUCT_EMULATIONS[rounding[site + '.prod']](x, w)
        return y if b is None else y + b
    return F.linear(_r(x, rounding, site + '.act'), _r(w, rounding, site + '.w'), b)


def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# ----------------------------------------------------------------------------------------------------------
# encoder  (timm VisionTransformer.forward_features, class_token=False, global_pool='')
# ----------------------------------------------------------------------------------------------------------

def vit_features(sd: dict, prefix: str, cfg: OracleConfig, images: Tensor, rounding: Optional[str] = None) -> Tensor:
    """timm VisionTransformer.forward_features with the parameters under `prefix` ('encoder.' for PARSeq, '' for ViTSTR).
    A class token is prepended (before the position embedding is added) iff `prefix + 'cls_token'` exists."""
    E, H = cfg.embed_dim, cfg.enc_num_heads
    hd = E // H
    B = images.shape[0]
    # PatchEmbed: Conv2d(k = stride = patch) then flatten(2).transpose(1, 2); token t = gy * grid_w + gx
    x = F.conv2d(_r(images, rounding, 'img'), _r(sd[prefix + 'patch_embed.proj.weight'], rounding, 'enc.w'),
                 sd[prefix + 'patch_embed.proj.bias'], stride=tuple(cfg.patch_size))
    x = x.flatten(2).transpose(1, 2)
    if prefix + 'cls_token' in sd:
        x = torch.cat([sd[prefix + 'cls_token'].expand(B, -1, -1), x], dim=1)
    x = x + sd[prefix + 'pos_embed']
    N = x.shape[1]
    for i in range(cfg.enc_depth):
        p = f'{prefix}blocks.{i}.'
        h = _ln(x, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], cfg.enc_ln_eps)
        qkv = _linear(h, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'], rounding, 'enc')
        qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        if rounding is None or (isinstance(rounding, dict) and 'enc.qkv' not in rounding and 'enc.p' not in rounding):
            a = F.scaled_dot_product_attention(q, k, v)  # default scale = hd ** -0.5, no mask
        else:
            # HIP bf16 path: q (pre-scaled by the exact power of two 0.125... generally hd**-0.5), k, v stored
            # bf16; scores/softmax fp32; un-normalised p rounded to bf16 for P.V; row sum kept in fp32.
            q, k, v = _r(q, rounding, 'enc.qkv'), _r(k, rounding, 'enc.qkv'), _r(v, rounding, 'enc.qkv')
            s = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
            m = s.amax(-1, keepdim=True)
            pexp = torch.exp(s - m)
            a = (_r(pexp, rounding, 'enc.p') @ v) / pexp.sum(-1, keepdim=True)
        a = a.transpose(1, 2).reshape(B, N, E)
        x = x + _linear(a, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'], rounding, 'enc')
        h = _ln(x, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], cfg.enc_ln_eps)
        h = F.gelu(_linear(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'], rounding, 'enc'))
        x = x + _linear(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'], rounding, 'enc')
    return _ln(x, sd[prefix + 'norm.weight'], sd[prefix + 'norm.bias'], cfg.enc_ln_eps)


def encode(sd: dict, cfg: OracleConfig, images: Tensor, rounding: Optional[str] = None) -> Tensor:
    """images [B,3,H,W] fp32 -> memory [B, N, E].  model.py:83-84, modules.py:163-165."""
    return vit_features(sd, 'encoder.', cfg, images, rounding)


# ----------------------------------------------------------------------------------------------------------
# decoder
# ----------------------------------------------------------------------------------------------------------

def token_embedding(sd: dict, cfg: OracleConfig, tokens: Tensor) -> Tensor:
    """modules.py:175-176: sqrt(E) * Embedding(tokens)."""
    return math.sqrt(cfg.embed_dim) * F.embedding(tokens, sd['text_embed.embedding.weight'])


def mha(sd: dict, prefix: str, num_heads: int, query: Tensor, key: Tensor,
        attn_mask: Optional[Tensor], key_padding_mask: Optional[Tensor], rounding: Optional[str]) -> Tensor:
    """nn.MultiheadAttention(batch_first=True) forward with query != key (so never the fused fast path) and
    need_weights=True: the explicit baddbmm / softmax / bmm branch.  Bool masks become additive -inf."""
    B, Lq, E = query.shape
    Lk = key.shape[1]
    hd = E // num_heads
    w, b = sd[prefix + 'in_proj_weight'], sd[prefix + 'in_proj_bias']
    # _in_projection_packed, q is not k branch: q from rows [0,E), k|v from rows [E,3E) in one linear
    q = _linear(query, w[:E], b[:E], rounding)
    kv = _linear(key, w[E:], b[E:], rounding)
    k, v = kv[..., :E], kv[..., E:]
    if rounding is not None:
        k, v = _r(_r(k, rounding, 'dec.kv'), rounding if isinstance(rounding, dict) else None, 'dec.k'), \
               _r(_r(v, rounding, 'dec.kv'), rounding if isinstance(rounding, dict) else None, 'dec.v')   # K/V are stored bf16 (memory K/V cache, content K/V table)
    q = q.reshape(B, Lq, num_heads, hd).transpose(1, 2).reshape(B * num_heads, Lq, hd)
    k = k.reshape(B, Lk, num_heads, hd).transpose(1, 2).reshape(B * num_heads, Lk, hd)
    v = v.reshape(B, Lk, num_heads, hd).transpose(1, 2).reshape(B * num_heads, Lk, hd)
    mask = None
    if attn_mask is not None:
        mask = torch.zeros(attn_mask.shape, dtype=q.dtype).masked_fill_(attn_mask, float('-inf')).unsqueeze(0)
    if key_padding_mask is not None:
        kpm = torch.zeros(key_padding_mask.shape, dtype=q.dtype).masked_fill_(key_padding_mask, float('-inf'))
        kpm = kpm.view(B, 1, 1, Lk).expand(-1, num_heads, -1, -1).reshape(B * num_heads, 1, Lk)
        mask = kpm if mask is None else mask + kpm
    q_scaled = q * math.sqrt(1.0 / float(hd))
    if mask is not None:
        s = torch.baddbmm(mask, q_scaled, k.transpose(-2, -1))
    else:
        s = torch.bmm(q_scaled, k.transpose(-2, -1))
    p = torch.softmax(s, dim=-1)
    o = torch.bmm(p, v)
    o = o.reshape(B, num_heads, Lq, hd).transpose(1, 2).reshape(B, Lq, E)
    return _linear(o, sd[prefix + 'out_proj.weight'], sd[prefix + 'out_proj.bias'], rounding)


def decoder_layer_query_stream(sd: dict, cfg: OracleConfig, query: Tensor, content: Tensor, memory: Tensor,
                               query_mask: Optional[Tensor], content_key_padding_mask: Optional[Tensor],
                               rounding: Optional[str]) -> Tensor:
    """DecoderLayer.forward with update_content=False (modules.py:81-98 -> forward_stream :55-79).
    dec_depth == 1, so the content stream is never updated (modules.py:120-123)."""
    p = 'decoder.layers.0.'
    eps = cfg.dec_ln_eps
    qn = _ln(query, sd[p + 'norm_q.weight'], sd[p + 'norm_q.bias'], eps)
    cn = _ln(content, sd[p + 'norm_c.weight'], sd[p + 'norm_c.bias'], eps)
    t = query + mha(sd, p + 'self_attn.', cfg.dec_num_heads, qn, cn, query_mask, content_key_padding_mask, rounding)
    # memory is NOT re-normed: it is the output of the ViT's final LayerNorm (modules.py:67)
    t = t + mha(sd, p + 'cross_attn.', cfg.dec_num_heads,
                _ln(t, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps), memory, None, None, rounding)
    h = _ln(t, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], eps)
    h = F.gelu(_linear(h, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'], rounding))
    t = t + _linear(h, sd[p + 'linear2.weight'], sd[p + 'linear2.bias'], rounding)
    return t


def decode(sd: dict, cfg: OracleConfig, tgt: Tensor, memory: Tensor, tgt_mask: Optional[Tensor] = None,
           tgt_padding_mask: Optional[Tensor] = None, tgt_query: Optional[Tensor] = None,
           tgt_query_mask: Optional[Tensor] = None, rounding: Optional[str] = None) -> Tensor:
    """model.py:86-103 followed by Decoder.forward (modules.py:110-125).  `tgt_mask` (the content mask) is
    accepted for signature parity but unused: with dec_depth == 1 the content stream is never updated."""
    assert cfg.dec_depth == 1, 'only dec_depth == 1 is supported (see SURVEY.md section 7, hard part 6)'
    N, L = tgt.shape
    pos_queries = sd['pos_queries']
    null_ctx = token_embedding(sd, cfg, tgt[:, :1])
    tgt_emb = pos_queries[:, :L - 1] + token_embedding(sd, cfg, tgt[:, 1:])
    content = torch.cat([null_ctx, tgt_emb], dim=1)
    if tgt_query is None:
        tgt_query = pos_queries[:, :L].expand(N, -1, -1)
    out = decoder_layer_query_stream(sd, cfg, tgt_query, content, memory, tgt_query_mask, tgt_padding_mask, rounding)
    return _ln(out, sd['decoder.norm.weight'], sd['decoder.norm.bias'], cfg.dec_ln_eps)


def head(sd: dict, x: Tensor, rounding: Optional[str] = None) -> Tensor:
    """model.py:63: Linear(E, num_tokens - 2)."""
    return _linear(x, sd['head.weight'], sd['head.bias'], rounding)


@dataclass
class Trace:
    memory: Optional[Tensor] = None
    ar_logits: Optional[Tensor] = None        # logits after the AR loop / NAR pass, before refinement
    ar_tokens: Optional[Tensor] = None        # tgt_in after the AR loop  [B, num_steps]
    refine_logits: list = field(default_factory=list)
    refine_tokens: list = field(default_factory=list)


def forward(sd: dict, cfg: OracleConfig, images: Tensor, max_length: Optional[int] = None, *,
            decode_ar: bool = True, refine_iters: int = 1, rounding: Optional[str] = None,
            trace: Optional[Trace] = None, teacher_tokens: Optional[Tensor] = None,
            teacher_refine_tokens: Optional[Sequence[Tensor]] = None) -> Tensor:
    """model.py:105-169.  Returns logits [B, L, num_tokens-2].

    `teacher_tokens` [B, num_steps] (optional, test aid): feed these as the AR context instead of this run's own
    argmax, and disable early exit, so a near-tie flip at one position cannot poison later positions when two
    arithmetic modes are compared (hard part 1-iii).  `teacher_refine_tokens[i]` does the
    same for refinement iteration i.
    """
    testing = max_length is None
    max_length = cfg.max_label_length if max_length is None else min(max_length, cfg.max_label_length)
    bs = images.shape[0]
    num_steps = max_length + 1
    memory = encode(sd, cfg, images, rounding)
    if trace is not None:
        trace.memory = memory
    pos_queries = sd['pos_queries'][:, :num_steps].expand(bs, -1, -1)
    tgt_mask = query_mask = torch.triu(torch.ones((num_steps, num_steps), dtype=torch.bool), 1)

    if decode_ar:
        tgt_in = torch.full((bs, num_steps), cfg.pad_id, dtype=torch.long)
        tgt_in[:, 0] = cfg.bos_id
        logits = []
        for i in range(num_steps):
            j = i + 1
            tgt_out = decode(sd, cfg, tgt_in[:, :j], memory, tgt_mask[:j, :j], tgt_query=pos_queries[:, i:j],
                             tgt_query_mask=query_mask[i:j, :j], rounding=rounding)
            p_i = head(sd, tgt_out, rounding)
            logits.append(p_i)
            if j < num_steps:
                if teacher_tokens is not None:
                    tgt_in[:, j] = teacher_tokens[:, j]
                else:
                    # p_i.squeeze() in the reference; reshape keeps batch-1 inputs well-formed
                    tgt_in[:, j] = p_i.reshape(bs, -1).argmax(-1)
                    if testing and (tgt_in == cfg.eos_id).any(dim=-1).all():
                        break
        logits = torch.cat(logits, dim=1)
        if trace is not None:
            trace.ar_tokens = tgt_in.clone()
    else:
        tgt_in = torch.full((bs, 1), cfg.bos_id, dtype=torch.long)
        tgt_out = decode(sd, cfg, tgt_in, memory, tgt_query=pos_queries, rounding=rounding)
        logits = head(sd, tgt_out, rounding)
    if trace is not None:
        trace.ar_logits = logits.clone()

    if refine_iters:
        # cloze mask: causal with everything to the right of i+1 unmasked (model.py:157); the reference edits the
        # aliased tensor in place, here it is a fresh tensor (tgt_mask is unused by a depth-1 decoder anyway)
        query_mask = query_mask.clone()
        query_mask[torch.triu(torch.ones(num_steps, num_steps, dtype=torch.bool), 2)] = False
        bos = torch.full((bs, 1), cfg.bos_id, dtype=torch.long)
        for it in range(refine_iters):
            if teacher_refine_tokens is not None:
                tgt_in = teacher_refine_tokens[it]
            else:
                tgt_in = torch.cat([bos, logits[:, :-1].argmax(-1)], dim=1)
            tgt_padding_mask = (tgt_in == cfg.eos_id).int().cumsum(-1) > 0
            tgt_out = decode(sd, cfg, tgt_in, memory, tgt_mask, tgt_padding_mask, pos_queries,
                             query_mask[:, :tgt_in.shape[1]], rounding=rounding)
            logits = head(sd, tgt_out, rounding)
            if trace is not None:
                trace.refine_tokens.append(tgt_in.clone())
                trace.refine_logits.append(logits.clone())
    return logits


def postprocess(logits: Tensor, eos_id: int = 0):
    """CPU restatement of the numeric half of `tokenizer.decode(logits.softmax(-1))` (row N1).

    Follows strhub/models/base.py:132-137 (`probs = logits.softmax(-1)`, `prob.prod()`), strhub/data/utils.py:90-91
    (`probs, ids = dist.max(-1)`) and :120-129 (`_filter`: characters stop before the first EOS, the probability list keeps
    the EOS probability).  Returns (ids [B, L] int64 for every position, lengths [B], probs [B, L], confidence [B]).
    """
    probs_all, ids_all = logits.float().softmax(-1).max(-1)
    B, L = ids_all.shape
    lengths = torch.full((B,), L, dtype=torch.long)
    conf = torch.empty(B, dtype=torch.float32)
    for b in range(B):
        row = ids_all[b].tolist()
        if eos_id in row:
            lengths[b] = row.index(eos_id)
        conf[b] = probs_all[b, :int(lengths[b]) + 1].prod()
    return ids_all, lengths, probs_all, conf


def normalize_u8(images_u8: Tensor) -> Tensor:
    """Row N2: the numeric tail of the reference's input transform (strhub/data/module.py:78-81) on uint8 CHW pixels —
    `T.ToTensor()` (`img.to(float32).div(255)`) followed by `T.Normalize(0.5, 0.5)` (`sub_(mean).div_(std)`)."""
    return images_u8.to(torch.float32).div(255).sub_(0.5).div_(0.5)


def validation_loss(logits: Tensor, targets: Tensor, pad_id: int):
    """strhub/models/base.py:199-200: (F.cross_entropy(logits.flatten(end_dim=1), targets.flatten(), ignore_index=pad_id),
    (targets != pad_id).sum()) — `targets` is tokenizer.encode(labels)[:, 1:]."""
    loss = F.cross_entropy(logits.float().flatten(end_dim=1), targets.flatten(), ignore_index=pad_id)
    return loss, (targets != pad_id).sum()


# ----------------------------------------------------------------------------------------------------------
# "next" row N3 (training step, forward half and — through autograd on this restatement — its gradients)
# ----------------------------------------------------------------------------------------------------------

def attn_masks_from_perm(perm: Tensor):
    """strhub/models/parseq/system.py:152-166 (`generate_attn_masks`).  `perm` lists sequence positions in the order
    they are generated (perm[0] == 0 is <bos>).  Position q may look at position k only when k comes earlier in `perm`;
    the query mask additionally hides q itself.  Returns bool (content_mask [sz-1, sz-1], query_mask [sz-1, sz-1]),
    True = masked.  Restated through the rank of every position instead of the reference's row-by-row fill."""
    sz = perm.shape[0]
    rank = torch.empty(sz, dtype=torch.long)
    rank[perm] = torch.arange(sz)
    later = rank.unsqueeze(0) > rank.unsqueeze(1)           # [q, k]: k is generated after q
    content_mask = later[:-1, :-1].clone()
    query_mask = (later | torch.eye(sz, dtype=torch.bool))[1:, :-1].clone()
    return content_mask, query_mask


def training_loss(sd: dict, cfg: OracleConfig, images: Tensor, tgt: Tensor, perms: Tensor, rounding: Optional[str] = None):
    """strhub/models/parseq/system.py:168-199 (`training_step`) with dropout off: one `encode`, one teacher-forced decode
    per permutation, cross-entropy weighted by the number of non-<pad> targets; after the second permutation the <eos>
    targets are dropped (:191-195).  `tgt` = tokenizer.encode(labels) [N, T+2]; `perms` = gen_tgt_perms(tgt) [K, T+2].
    Returns (loss, per-permutation losses [K], per-permutation target counts [K])."""
    memory = encode(sd, cfg, images, rounding)
    tgt_in, tgt_out = tgt[:, :-1], tgt[:, 1:]
    tgt_padding_mask = (tgt_in == cfg.pad_id) | (tgt_in == cfg.eos_id)
    total, numel, per_perm, counts = 0.0, 0, [], []
    n = int((tgt_out != cfg.pad_id).sum())
    for i, perm in enumerate(perms):
        _, query_mask = attn_masks_from_perm(perm)
        out = decode(sd, cfg, tgt_in, memory, None, tgt_padding_mask, tgt_query_mask=query_mask, rounding=rounding)
        logits = head(sd, out, rounding).flatten(end_dim=1)
        ce = F.cross_entropy(logits.float(), tgt_out.flatten(), ignore_index=cfg.pad_id)
        total = total + n * ce
        numel += n
        per_perm.append(ce.detach())
        counts.append(n)
        if i == 1:
            tgt_out = torch.where(tgt_out == cfg.eos_id, cfg.pad_id, tgt_out)
            n = int((tgt_out != cfg.pad_id).sum())
    return total / numel, torch.stack(per_perm), torch.tensor(counts)
