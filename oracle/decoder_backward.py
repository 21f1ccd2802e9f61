"""Hand-derived backward of the K-permutation training loss, decoder side.   *** TEST INFRASTRUCTURE ***

Row N3.  The reference gets its gradients from autograd (`loss.backward()` after strhub/models/parseq/system.py:168-199);
a HIP implementation has to spell the chain rule out.  This module spells it out ONCE on the CPU, operator by operator and
in the order the device code launches its kernels (parseq_amd/csrc/train_ops.h, `parseq_train_decoder`), so that

  * the derivation itself is checked against autograd through the oracle (tests/test_training.py) with no GPU involved, and
  * every device kernel has a one-line CPU counterpart with identical semantics to be compared with on the GPU.

Nothing here is a product path.  Operators (all fp32, row-major):

  ln(x, w, b)                    LayerNorm over the last axis (eps = 1e-5, modules.py:38-41 via nn.LayerNorm defaults)
  ln_bwd(x, w, dy)               -> dx, dw, db                       (statistics recomputed from x)
  attn(q, k, v, mask)            soft-max attention per (batch, head); q [B|1, H, Lq, d], k / v [B, H, Lk, d], mask [B|1, Lq, Lk] bool
  attn_bwd(q, k, v, mask, do)    -> dq, dk, dv                       (probabilities recomputed)
  gelu_bwd(pre, dact)            exact erf GELU (F.gelu default, modules.py:43,77)
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

EPS = 1e-5
HD = 32          # decoder head width (embed_dim / dec_num_heads, configs/model/parseq.yaml:11-12)

# Arithmetic of every matrix product below: None = exact fp32 (the device's first version, and the parity gate), 'bf16' = both
# operands rounded to bfloat16, fp32 accumulate — what a matrix-core bf16 training step computes (BASELINE configs[4]); used to
# size the gradient error that rounding alone introduces, i.e. the tolerance a bf16 device path can be held to.
_ROUNDING = [None]


class rounding:
    """`with rounding('bf16'): ...` — switch the arithmetic of `mm` inside the block."""

    def __init__(self, mode):
        assert mode in (None, 'bf16')
        self.mode = mode

    def __enter__(self):
        self.prev, _ROUNDING[0] = _ROUNDING[0], self.mode

    def __exit__(self, *exc):
        _ROUNDING[0] = self.prev


def mm(a, b):
    if _ROUNDING[0] == 'bf16':
        return a.to(torch.bfloat16).float() @ b.to(torch.bfloat16).float()
    return a @ b


def ln(x, w, b, eps=EPS):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def ln_bwd(x, w, dy, eps=EPS):
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((x - mu) ** 2).mean(-1, keepdim=True) + eps)
    xhat = (x - mu) * rstd
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True))
    flat = lambda t: t.reshape(-1, t.shape[-1])
    return dx, flat(dy * xhat).sum(0), flat(dy).sum(0)


def split_heads(x, B, L, hd=HD):     # [B * L, E] -> [B, H, L, d]
    return x.reshape(B, L, -1, hd).transpose(1, 2)


def merge_heads(x):                  # [B, H, L, d] -> [B * L, E]
    B, H, L, d = x.shape
    return x.transpose(1, 2).reshape(B * L, H * d)


def _mix(x):
    """The 32-bit integer mixer of parseq_amd/csrc/train_ops.h:drop_mix, on int64 tensors holding uint32 values."""
    m = 0xFFFFFFFF
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & m
    x = x ^ (x >> 15); x = (x * 0x846ca68b) & m
    return x ^ (x >> 16)


class Dropout:
    """train_ops.h:DropSpec / drop_factor restated: element idx of site is kept iff hash(seed, site, idx) >= p * 2^32."""

    def __init__(self, p: float = 0.0, seed: int = 0):
        self.p = p
        self.seed_lo, self.seed_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        import numpy as np
        p32 = np.float32(p)                          # the C ABI takes the probability as a float
        self.thresh = int(float(p32) * 4294967296.0) if p > 0 else 0
        self.scale = float(np.float32(1.0) / (np.float32(1.0) - p32)) if p > 0 else 1.0

    def factor(self, site: int, shape) -> torch.Tensor:
        """Multiplier (0 or 1 / (1 - p)) of every element of a contiguous tensor of `shape` (element index = flat index)."""
        n = 1
        for d in shape:
            n *= d
        if self.thresh == 0:
            return torch.ones(shape)
        m = 0xFFFFFFFF
        idx = torch.arange(n, dtype=torch.int64)
        h = _mix((idx & m) ^ self.seed_lo)
        h = _mix((h + (idx >> 32) * 0x9e3779b9 + site * 0x85ebca6b + self.seed_hi) & m)
        return torch.where(h >= self.thresh, self.scale, 0.0).to(torch.float32).view(shape)


# dropout sites of one permutation pass (site id = 8 * permutation + these), in launch order
S_CONTENT, S_QUERY, S_SA_PROB, S_SA_OUT, S_CA_PROB, S_CA_OUT, S_FF_HIDDEN, S_FF_OUT = range(8)


def attn_probs(q, k, mask):
    s = mm(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1), float('-inf'))
    return torch.softmax(s, dim=-1)


def attn(q, k, v, mask, pf=None):
    """`pf`: dropout multipliers of the probabilities [B, H, Lq, Lk] (modules.py:33-34: nn.MultiheadAttention(dropout=...))."""
    p = attn_probs(q, k, mask)
    return mm(p if pf is None else p * pf, v)


def attn_bwd(q, k, v, mask, do, pf=None):
    p = attn_probs(q, k, mask)
    pd = p if pf is None else p * pf
    dv = mm(pd.transpose(-1, -2), do)
    dp = mm(do, v.transpose(-1, -2))
    if pf is not None:
        dp = dp * pf
    ds = p * (dp - (dp * p).sum(-1, keepdim=True)) * (1.0 / math.sqrt(q.shape[-1]))
    return mm(ds, k), mm(ds.transpose(-1, -2), q), dv          # dq (per batch even when q is shared), dk, dv


def gelu_bwd(pre, dact):
    cdf = 0.5 * (1.0 + torch.erf(pre * (1.0 / math.sqrt(2.0))))
    pdf = torch.exp(-0.5 * pre * pre) * (1.0 / math.sqrt(2.0 * math.pi))
    return dact * (cdf + pre * pdf)


def loss_and_grads(sd: dict, cfg, memory: torch.Tensor, tgt: torch.Tensor, perms: torch.Tensor, masks_fn, trace: dict = None,
                   dropout: Dropout = None):
    """Loss of system.py:168-199 given the encoder output, with its gradient w.r.t. every decoder-side parameter (`decoder.*`,
    `head.*`, `text_embed.*`, `pos_queries`) and w.r.t. `memory`.  `masks_fn(perm)` -> (content_mask, query_mask).  `dropout`
    (default: off) supplies the masks of the eight dropout sites of every permutation pass — the embeddings and queries are
    dropped afresh in each `decode` call (model.py:99-102), so nothing but the memory's K / V is shared between passes.
    Returns (loss, per-permutation losses, grads dict, dmemory).  `trace`, if given, receives the intermediates of the LAST
    permutation and the cross-permutation accumulators under the names of `parseq_train_decoder_workspace_offset`."""
    E, S = cfg.embed_dim, memory.shape[1]
    p = 'decoder.layers.0.'
    W = lambda k: sd[k].detach()
    B = tgt.shape[0]
    tgt_in, tgt_out = tgt[:, :-1], tgt[:, 1:]
    L = tgt_in.shape[1]
    M = B * L
    H = E // HD
    drop = dropout if dropout is not None else Dropout()
    pad = (tgt_in == cfg.pad_id) | (tgt_in == cfg.eos_id)                                # [B, L]
    grads = {k: torch.zeros_like(v) for k, v in sd.items() if not k.startswith('encoder.')}

    def acc_linear(wkey, bkey, rows, x, dy):
        """dW[rows] += dy^T x ; db[rows] += colsum(dy) ; returns dx = dy W[rows]"""
        grads[wkey][rows] += mm(dy.t(), x)
        grads[bkey][rows] += dy.sum(0)
        return mm(dy, W(wkey)[rows])

    def acc_ln(key, x, dy):
        dx, dw, db = ln_bwd(x, W(key + '.weight'), dy)
        grads[key + '.weight'] += dw; grads[key + '.bias'] += db
        return dx

    # ---- shared by all permutations: the undropped content rows and the memory's K / V ---------------------------------
    pq = W('pos_queries')[0, :L]                                                          # [L, E]
    content0 = math.sqrt(E) * W('text_embed.embedding.weight')[tgt_in]                     # [B, L, E]
    content0[:, 1:] += pq[:L - 1]
    content0 = content0.reshape(M, E)
    sa_w, sa_b = W(p + 'self_attn.in_proj_weight'), W(p + 'self_attn.in_proj_bias')
    ca_w, ca_b = W(p + 'cross_attn.in_proj_weight'), W(p + 'cross_attn.in_proj_bias')
    mem2 = memory.detach().reshape(B * S, E)
    kvm = mm(mem2, ca_w[E:].t()) + ca_b[E:]                                                  # [B * S, 2E]
    km, vm = split_heads(kvm[:, :E], B, S), split_heads(kvm[:, E:], B, S)

    n_first = int((tgt_out != cfg.pad_id).sum())
    tgt_late = torch.where(tgt_out == cfg.eos_id, cfg.pad_id, tgt_out)
    n_late = int((tgt_late != cfg.pad_id).sum())
    K = len(perms)
    total = n_first * min(K, 2) + n_late * max(K - 2, 0)

    d_content = torch.zeros(M, E)
    d_kvm = torch.zeros(B * S, 2 * E)
    d_pq = torch.zeros(L, E)
    losses, weighted = [], 0.0
    every = slice(None)
    for i, perm in enumerate(perms):
        site = lambda s: 8 * i + s
        qmask = masks_fn(perm)[1]                                                         # [L, L]
        sa_mask = qmask.unsqueeze(0) | pad.unsqueeze(1)                                   # [B, L, L]
        targets = (tgt_out if i < 2 else tgt_late).reshape(M)
        f_content, f_query = drop.factor(site(S_CONTENT), (M, E)), drop.factor(site(S_QUERY), (M, E))
        f_sa, f_ca = drop.factor(site(S_SA_PROB), (B, H, L, L)), drop.factor(site(S_CA_PROB), (B, H, L, S))
        f_sa_out, f_ca_out = drop.factor(site(S_SA_OUT), (M, E)), drop.factor(site(S_CA_OUT), (M, E))
        f_hidden, f_ff_out = drop.factor(site(S_FF_HIDDEN), (M, 4 * E)), drop.factor(site(S_FF_OUT), (M, E))
        # forward (model.py:95-103, modules.py:55-98)
        content = content0 * f_content
        cn = ln(content, W(p + 'norm_c.weight'), W(p + 'norm_c.bias'))
        kvc = mm(cn, sa_w[E:].t()) + sa_b[E:]                                                # [M, 2E]
        kc, vc = split_heads(kvc[:, :E], B, L), split_heads(kvc[:, E:], B, L)
        qd = pq.repeat(B, 1) * f_query                                                    # the query stream's input, per image
        qn = ln(qd, W(p + 'norm_q.weight'), W(p + 'norm_q.bias'))
        qsa = mm(qn, sa_w[:E].t()) + sa_b[:E]
        q_sa = split_heads(qsa, B, L)
        sa_o = merge_heads(attn(q_sa, kc, vc, sa_mask, f_sa))                             # [M, E]
        t1 = qd + (mm(sa_o, W(p + 'self_attn.out_proj.weight').t()) + W(p + 'self_attn.out_proj.bias')) * f_sa_out
        n1 = ln(t1, W(p + 'norm1.weight'), W(p + 'norm1.bias'))
        q2 = split_heads(mm(n1, ca_w[:E].t()) + ca_b[:E], B, L)
        ca_o = merge_heads(attn(q2, km, vm, None, f_ca))
        t2 = t1 + (mm(ca_o, W(p + 'cross_attn.out_proj.weight').t()) + W(p + 'cross_attn.out_proj.bias')) * f_ca_out
        n2 = ln(t2, W(p + 'norm2.weight'), W(p + 'norm2.bias'))
        hpre = mm(n2, W(p + 'linear1.weight').t()) + W(p + 'linear1.bias')
        hact = F.gelu(hpre) * f_hidden
        t3 = t2 + (mm(hact, W(p + 'linear2.weight').t()) + W(p + 'linear2.bias')) * f_ff_out
        out = ln(t3, W('decoder.norm.weight'), W('decoder.norm.bias'))
        logits = mm(out, W('head.weight').t()) + W('head.bias')
        keep = targets != cfg.pad_id
        logp = torch.log_softmax(logits, -1)
        row_loss = -logp[keep, targets[keep]]
        losses.append(row_loss.mean())
        weighted = weighted + row_loss.sum()
        # backward: d(total loss)/d logits = (softmax - onehot) / total on the kept rows, 0 elsewhere
        dlogits = torch.zeros_like(logits)
        dlogits[keep] = torch.softmax(logits[keep], -1)
        dlogits[keep, targets[keep]] -= 1.0
        dlogits /= total
        dout = acc_linear('head.weight', 'head.bias', every, out, dlogits)
        dt3 = acc_ln('decoder.norm', t3, dout)
        dhact = acc_linear(p + 'linear2.weight', p + 'linear2.bias', every, hact, dt3 * f_ff_out)
        dn2 = acc_linear(p + 'linear1.weight', p + 'linear1.bias', every, n2, gelu_bwd(hpre, dhact * f_hidden))
        dt2 = dt3 + acc_ln(p + 'norm2', t2, dn2)
        dca_o = acc_linear(p + 'cross_attn.out_proj.weight', p + 'cross_attn.out_proj.bias', every, ca_o, dt2 * f_ca_out)
        dq2, dkm, dvm = attn_bwd(q2, km, vm, None, split_heads(dca_o, B, L), f_ca)
        d_kvm[:, :E] += merge_heads(dkm); d_kvm[:, E:] += merge_heads(dvm)
        dn1 = acc_linear(p + 'cross_attn.in_proj_weight', p + 'cross_attn.in_proj_bias', slice(0, E), n1, merge_heads(dq2))
        dt1 = dt2 + acc_ln(p + 'norm1', t1, dn1)
        dsa_o = acc_linear(p + 'self_attn.out_proj.weight', p + 'self_attn.out_proj.bias', every, sa_o, dt1 * f_sa_out)
        dq, dk, dv = attn_bwd(q_sa, kc, vc, sa_mask, split_heads(dsa_o, B, L), f_sa)
        dqn = acc_linear(p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias', slice(0, E), qn, merge_heads(dq))
        dqd = dt1 + acc_ln(p + 'norm_q', qd, dqn)                                         # residual input + norm_q input
        d_pq += (dqd * f_query).view(B, L, E).sum(0)
        d_kvc = torch.cat([merge_heads(dk), merge_heads(dv)], dim=1)
        dcn = acc_linear(p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias', slice(E, 3 * E), cn, d_kvc)
        d_content += acc_ln(p + 'norm_c', content, dcn) * f_content
        if trace is not None and i == K - 1:
            trace.update(content=content, cn=cn, kvc=kvc, qd=qd, qn=qn, qsa=qsa, kvm=kvm, sa_o=sa_o, t1=t1, n1=n1, q2=merge_heads(q2),
                         ca_o=ca_o, t2=t2, n2=n2, hpre=hpre, hact=hact, t3=t3, out=out, dlogits=dlogits,
                         d_kvc=d_kvc, d_kvm=d_kvm, d_content=d_content)

    # ---- what every permutation shares, once -----------------------------------------------------------------------------
    d_content = d_content.view(B, L, E)
    d_pq[:L - 1] += d_content[:, 1:].sum(0)
    grads['text_embed.embedding.weight'].index_add_(0, tgt_in.reshape(M), math.sqrt(E) * d_content.reshape(M, E))
    dmem = acc_linear(p + 'cross_attn.in_proj_weight', p + 'cross_attn.in_proj_bias', slice(E, 3 * E), mem2, d_kvm)
    grads['pos_queries'][0, :L] += d_pq
    return weighted / total, torch.stack(losses), grads, dmem.view(B, S, E)
