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


def attn_probs(q, k, mask):
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1), float('-inf'))
    return torch.softmax(s, dim=-1)


def attn(q, k, v, mask):
    return attn_probs(q, k, mask) @ v


def attn_bwd(q, k, v, mask, do):
    p = attn_probs(q, k, mask)
    dv = p.transpose(-1, -2) @ do
    dp = do @ v.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True)) * (1.0 / math.sqrt(q.shape[-1]))
    return ds @ k, ds.transpose(-1, -2) @ q, dv          # dq (per batch even when q is shared), dk, dv


def gelu_bwd(pre, dact):
    cdf = 0.5 * (1.0 + torch.erf(pre * (1.0 / math.sqrt(2.0))))
    pdf = torch.exp(-0.5 * pre * pre) * (1.0 / math.sqrt(2.0 * math.pi))
    return dact * (cdf + pre * pdf)


def loss_and_grads(sd: dict, cfg, memory: torch.Tensor, tgt: torch.Tensor, perms: torch.Tensor, masks_fn, trace: dict = None):
    """Loss of system.py:168-199 (dropout off) given the encoder output, with its gradient w.r.t. every decoder-side parameter
    (`decoder.*`, `head.*`, `text_embed.*`, `pos_queries`) and w.r.t. `memory`.  `masks_fn(perm)` -> (content_mask, query_mask).
    Returns (loss, per-permutation losses, grads dict, dmemory).  `trace`, if given, receives the intermediates of the LAST
    permutation and the cross-permutation accumulators under the names of `parseq_train_decoder_workspace_offset`."""
    E, S = cfg.embed_dim, memory.shape[1]
    p = 'decoder.layers.0.'
    W = lambda k: sd[k].detach()
    B = tgt.shape[0]
    tgt_in, tgt_out = tgt[:, :-1], tgt[:, 1:]
    L = tgt_in.shape[1]
    M = B * L
    pad = (tgt_in == cfg.pad_id) | (tgt_in == cfg.eos_id)                                # [B, L]
    grads = {k: torch.zeros_like(v) for k, v in sd.items() if not k.startswith('encoder.')}

    def acc_linear(wkey, bkey, rows, x, dy):
        """dW[rows] += dy^T x ; db[rows] += colsum(dy) ; returns dx = dy W[rows]"""
        grads[wkey][rows] += dy.t() @ x
        grads[bkey][rows] += dy.sum(0)
        return dy @ W(wkey)[rows]

    # ---- shared by all permutations ---------------------------------------------------------------------------------
    pq = W('pos_queries')[0, :L]                                                          # [L, E]
    emb = math.sqrt(E) * W('text_embed.embedding.weight')[tgt_in]                          # [B, L, E]
    content = emb.clone()
    content[:, 1:] += pq[:L - 1]
    content = content.reshape(M, E)
    cn = ln(content, W(p + 'norm_c.weight'), W(p + 'norm_c.bias'))
    sa_w, sa_b = W(p + 'self_attn.in_proj_weight'), W(p + 'self_attn.in_proj_bias')
    kvc = cn @ sa_w[E:].t() + sa_b[E:]                                                    # [M, 2E]
    kc, vc = split_heads(kvc[:, :E], B, L), split_heads(kvc[:, E:], B, L)
    qn = ln(pq, W(p + 'norm_q.weight'), W(p + 'norm_q.bias'))                              # [L, E] — the same for every image
    q_sa = split_heads(qn @ sa_w[:E].t() + sa_b[:E], 1, L)                                # [1, H, L, d]
    ca_w, ca_b = W(p + 'cross_attn.in_proj_weight'), W(p + 'cross_attn.in_proj_bias')
    mem2 = memory.detach().reshape(B * S, E)
    kvm = mem2 @ ca_w[E:].t() + ca_b[E:]                                                  # [B * S, 2E]
    km, vm = split_heads(kvm[:, :E], B, S), split_heads(kvm[:, E:], B, S)

    n_first = int((tgt_out != cfg.pad_id).sum())
    tgt_late = torch.where(tgt_out == cfg.eos_id, cfg.pad_id, tgt_out)
    n_late = int((tgt_late != cfg.pad_id).sum())
    K = len(perms)
    total = n_first * min(K, 2) + n_late * max(K - 2, 0)

    d_qsa = torch.zeros(L, E)
    d_kvc = torch.zeros(M, 2 * E)
    d_kvm = torch.zeros(B * S, 2 * E)
    d_pq = torch.zeros(L, E)
    losses, weighted = [], 0.0
    for i, perm in enumerate(perms):
        qmask = masks_fn(perm)[1]                                                         # [L, L]
        sa_mask = qmask.unsqueeze(0) | pad.unsqueeze(1)                                   # [B, L, L]
        targets = (tgt_out if i < 2 else tgt_late).reshape(M)
        # forward
        sa_o = merge_heads(attn(q_sa, kc, vc, sa_mask))                                   # [M, E]
        t1 = pq.repeat(B, 1) + sa_o @ W(p + 'self_attn.out_proj.weight').t() + W(p + 'self_attn.out_proj.bias')
        n1 = ln(t1, W(p + 'norm1.weight'), W(p + 'norm1.bias'))
        q2 = split_heads(n1 @ ca_w[:E].t() + ca_b[:E], B, L)
        ca_o = merge_heads(attn(q2, km, vm, None))
        t2 = t1 + ca_o @ W(p + 'cross_attn.out_proj.weight').t() + W(p + 'cross_attn.out_proj.bias')
        n2 = ln(t2, W(p + 'norm2.weight'), W(p + 'norm2.bias'))
        hpre = n2 @ W(p + 'linear1.weight').t() + W(p + 'linear1.bias')
        hact = F.gelu(hpre)
        t3 = t2 + hact @ W(p + 'linear2.weight').t() + W(p + 'linear2.bias')
        out = ln(t3, W('decoder.norm.weight'), W('decoder.norm.bias'))
        logits = out @ W('head.weight').t() + W('head.bias')
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
        every = slice(None)
        dout = acc_linear('head.weight', 'head.bias', every, out, dlogits)
        dt3, dw, db = ln_bwd(t3, W('decoder.norm.weight'), dout)
        grads['decoder.norm.weight'] += dw; grads['decoder.norm.bias'] += db
        dhact = acc_linear(p + 'linear2.weight', p + 'linear2.bias', every, hact, dt3)
        dn2 = acc_linear(p + 'linear1.weight', p + 'linear1.bias', every, n2, gelu_bwd(hpre, dhact))
        dx, dw, db = ln_bwd(t2, W(p + 'norm2.weight'), dn2)
        grads[p + 'norm2.weight'] += dw; grads[p + 'norm2.bias'] += db
        dt2 = dt3 + dx
        dca_o = acc_linear(p + 'cross_attn.out_proj.weight', p + 'cross_attn.out_proj.bias', every, ca_o, dt2)
        dq2, dkm, dvm = attn_bwd(q2, km, vm, None, split_heads(dca_o, B, L))
        d_kvm[:, :E] += merge_heads(dkm); d_kvm[:, E:] += merge_heads(dvm)
        dn1 = acc_linear(p + 'cross_attn.in_proj_weight', p + 'cross_attn.in_proj_bias', slice(0, E), n1, merge_heads(dq2))
        dx, dw, db = ln_bwd(t1, W(p + 'norm1.weight'), dn1)
        grads[p + 'norm1.weight'] += dw; grads[p + 'norm1.bias'] += db
        dt1 = dt2 + dx
        dsa_o = acc_linear(p + 'self_attn.out_proj.weight', p + 'self_attn.out_proj.bias', every, sa_o, dt1)
        d_pq += dt1.view(B, L, E).sum(0)                                                  # the query stream's residual input
        dq, dk, dv = attn_bwd(q_sa, kc, vc, sa_mask, split_heads(dsa_o, B, L))
        d_qsa += merge_heads(dq).view(B, L, E).sum(0)
        d_kvc[:, :E] += merge_heads(dk); d_kvc[:, E:] += merge_heads(dv)
        if trace is not None and i == K - 1:
            trace.update(content=content, cn=cn, kvc=kvc, qn=qn, qsa=merge_heads(q_sa), kvm=kvm, sa_o=sa_o, t1=t1, n1=n1, q2=merge_heads(q2),
                         ca_o=ca_o, t2=t2, n2=n2, hpre=hpre, hact=hact, t3=t3, out=out, dlogits=dlogits,
                         d_kvc=d_kvc, d_kvm=d_kvm, d_qsa=d_qsa)

    # ---- the shared prefix, once ----------------------------------------------------------------------------------------
    dqn = acc_linear(p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias', slice(0, E), qn, d_qsa)
    dx, dw, db = ln_bwd(pq, W(p + 'norm_q.weight'), dqn)
    grads[p + 'norm_q.weight'] += dw; grads[p + 'norm_q.bias'] += db
    d_pq += dx
    dcn = acc_linear(p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias', slice(E, 3 * E), cn, d_kvc)
    dcontent, dw, db = ln_bwd(content, W(p + 'norm_c.weight'), dcn)
    grads[p + 'norm_c.weight'] += dw; grads[p + 'norm_c.bias'] += db
    dcontent = dcontent.view(B, L, E)
    d_pq[:L - 1] += dcontent[:, 1:].sum(0)
    grads['text_embed.embedding.weight'].index_add_(0, tgt_in.reshape(M), math.sqrt(E) * dcontent.reshape(M, E))
    dmem = acc_linear(p + 'cross_attn.in_proj_weight', p + 'cross_attn.in_proj_bias', slice(E, 3 * E), mem2, d_kvm)
    grads['pos_queries'][0, :L] += d_pq
    return weighted / total, torch.stack(losses), grads, dmem.view(B, S, E)
