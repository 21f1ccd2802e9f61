"""Hand-derived backward of the ViT encoder (timm VisionTransformer blocks as PARSeq uses them).   *** TEST INFRASTRUCTURE ***

Row N3, second half: the gradient that arrives at the encoder output (`dmemory`, produced by the decoder's backward) is pushed
through encoder.norm, the twelve pre-norm blocks and the patch embedding, in the order the device code launches its kernels
(`parseq_train_encoder_forward` / `parseq_train_encoder_backward`).  Same operators as oracle/decoder_backward.py; the forward
restates oracle/parseq_oracle.py:vit_features (timm==0.9.16 semantics, see oracle/timm_standin.py) keeping what the backward
needs: per block the residual stream before it (x), qkv, the attention output, the stream after the attention residual
(x_mid) and the fc1 pre-activation; LayerNorm outputs and GELU are recomputed.  Checked against the reference's gradients
(tests/golden/parseq_train.*) in tests/test_training.py.  Nothing here is a product path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .decoder_backward import attn, attn_bwd, gelu_bwd, ln, ln_bwd, merge_heads, mm, split_heads


def patches_of(images: torch.Tensor, ph: int, pw: int) -> torch.Tensor:
    """[B, 3, H, W] -> [B * tokens, 3 * ph * pw]: row = (b, gy, gx), column = (c, ky, kx) — the Conv2d weight's own flattening."""
    B, C, H, W = images.shape
    x = images.reshape(B, C, H // ph, ph, W // pw, pw).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B * (H // ph) * (W // pw), C * ph * pw)


def forward(sd: dict, cfg, images: torch.Tensor):
    """-> (memory [B, S, E], saved activations)."""
    E, H = cfg.embed_dim, cfg.enc_num_heads
    hd, eps = E // H, cfg.enc_ln_eps
    B = images.shape[0]
    ph, pw = cfg.patch_size
    W = lambda k: sd['encoder.' + k].detach()
    patches = patches_of(images, ph, pw)
    S = patches.shape[0] // B
    x = mm(patches, W('patch_embed.proj.weight').reshape(E, -1).t()) + W('patch_embed.proj.bias') + W('pos_embed')[0].repeat(B, 1)
    saved = {'patches': patches, 'blocks': []}
    for i in range(cfg.enc_depth):
        p = f'blocks.{i}.'
        n1 = ln(x, W(p + 'norm1.weight'), W(p + 'norm1.bias'), eps)
        qkv = mm(n1, W(p + 'attn.qkv.weight').t()) + W(p + 'attn.qkv.bias')
        q, k, v = (split_heads(qkv[:, j * E:(j + 1) * E], B, S, hd) for j in range(3))
        ao = merge_heads(attn(q, k, v, None))
        x_mid = x + mm(ao, W(p + 'attn.proj.weight').t()) + W(p + 'attn.proj.bias')
        n2 = ln(x_mid, W(p + 'norm2.weight'), W(p + 'norm2.bias'), eps)
        hpre = mm(n2, W(p + 'mlp.fc1.weight').t()) + W(p + 'mlp.fc1.bias')
        x_out = x_mid + mm(F.gelu(hpre), W(p + 'mlp.fc2.weight').t()) + W(p + 'mlp.fc2.bias')
        saved['blocks'].append({'x': x, 'qkv': qkv, 'ao': ao, 'x_mid': x_mid, 'hpre': hpre})
        x = x_out
    saved['x_last'] = x
    return ln(x, W('norm.weight'), W('norm.bias'), eps).view(B, S, E), saved


def backward(sd: dict, cfg, saved: dict, dmemory: torch.Tensor) -> dict:
    """Gradients of every `encoder.*` parameter given d loss / d memory."""
    E, H = cfg.embed_dim, cfg.enc_num_heads
    hd, eps = E // H, cfg.enc_ln_eps
    B, S = dmemory.shape[0], dmemory.shape[1]
    W = lambda k: sd['encoder.' + k].detach()
    grads = {k: torch.zeros_like(v) for k, v in sd.items() if k.startswith('encoder.')}
    G = lambda k: grads['encoder.' + k]

    def lin_bwd(wkey, bkey, x, dy):
        G(wkey).view(dy.shape[1], -1).add_(mm(dy.t(), x))
        G(bkey).add_(dy.sum(0))
        return mm(dy, W(wkey).reshape(dy.shape[1], -1))

    dx, dw, db = ln_bwd(saved['x_last'], W('norm.weight'), dmemory.reshape(B * S, E), eps)
    G('norm.weight').add_(dw); G('norm.bias').add_(db)
    for i in reversed(range(cfg.enc_depth)):
        p = f'blocks.{i}.'
        a = saved['blocks'][i]
        hact = F.gelu(a['hpre'])
        dhact = lin_bwd(p + 'mlp.fc2.weight', p + 'mlp.fc2.bias', hact, dx)
        n2 = ln(a['x_mid'], W(p + 'norm2.weight'), W(p + 'norm2.bias'), eps)
        dn2 = lin_bwd(p + 'mlp.fc1.weight', p + 'mlp.fc1.bias', n2, gelu_bwd(a['hpre'], dhact))
        d, dw, db = ln_bwd(a['x_mid'], W(p + 'norm2.weight'), dn2, eps)
        G(p + 'norm2.weight').add_(dw); G(p + 'norm2.bias').add_(db)
        dx_mid = dx + d
        dao = lin_bwd(p + 'attn.proj.weight', p + 'attn.proj.bias', a['ao'], dx_mid)
        q, k, v = (split_heads(a['qkv'][:, j * E:(j + 1) * E], B, S, hd) for j in range(3))
        dq, dk, dv = attn_bwd(q, k, v, None, split_heads(dao, B, S, hd))
        dqkv = torch.cat([merge_heads(dq), merge_heads(dk), merge_heads(dv)], dim=1)
        n1 = ln(a['x'], W(p + 'norm1.weight'), W(p + 'norm1.bias'), eps)
        dn1 = lin_bwd(p + 'attn.qkv.weight', p + 'attn.qkv.bias', n1, dqkv)
        d, dw, db = ln_bwd(a['x'], W(p + 'norm1.weight'), dn1, eps)
        G(p + 'norm1.weight').add_(dw); G(p + 'norm1.bias').add_(db)
        dx = dx_mid + d
    G('pos_embed').add_(dx.view(B, S, E).sum(0))
    lin_bwd('patch_embed.proj.weight', 'patch_embed.proj.bias', saved['patches'], dx)
    return grads
