"""Minimal in-memory stand-in for the three `timm` symbols the reference PARSeq / ViTSTR models import.

TEST INFRASTRUCTURE ONLY.  Used by `oracle/make_golden.py` (in the build container, where
`/root/reference` exists) so that the reference's *own* `strhub/models/parseq/model.py` and
`modules.py` can be executed unmodified to mint golden vectors.  Nothing under `parseq_amd/`
imports this file.

The reference pins `timm==0.9.16` (reference `requirements/core.txt:32`) and uses exactly
  * `timm.models.helpers.named_apply`                      (reference `strhub/models/parseq/model.py:23,70`)
  * `timm.models.vision_transformer.PatchEmbed`            (reference `strhub/models/parseq/modules.py:24,143`)
  * `timm.models.vision_transformer.VisionTransformer`     (reference `strhub/models/parseq/modules.py:24,128-165`)
timm is NOT vendored under /root/reference and is not installed here, so this is a restatement of
its published ViT semantics for the constructor arguments the reference passes
(`num_classes=0, global_pool='', class_token=False, qkv_bias=True`, all drop rates 0):
pre-LN blocks, LayerNorm eps 1e-6, exact-erf GELU, fused qkv Linear laid out [3, heads, head_dim],
softmax(q k^T * head_dim**-0.5) v, learned `pos_embed[1, N, E]` added after the patch projection.
Parity at this boundary is therefore "unpinned" (see DESIGN.md section 3); what constrains it is
listed there (parameter counts, state-dict keys, and an independent cross-check of the block against
`transformers`' ViTLayer in tests/test_oracle.py).
"""
from __future__ import annotations

import sys
import types
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F


def named_apply(fn, module: nn.Module, name: str = '', depth_first: bool = True, include_root: bool = False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child in module.named_children():
        child_full = '.'.join((name, child_name)) if name else child_name
        named_apply(fn=fn, module=child, name=child_full, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True, bias=True):
        super().__init__()
        self.img_size = _pair(img_size)
        self.patch_size = _pair(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)  # [B, E, gh, gw] -> [B, gh*gw, E]
        return self.norm(x)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        x = F.scaled_dot_product_attention(q, k, v)
        x = x.transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class VisionTransformer(nn.Module):
    """Two configurations are restated (timm 0.9.16 `vision_transformer.py`):
      * the PARSeq encoder: num_classes=0, global_pool='', class_token=False (no head, no class token);
      * ViTSTR (strhub/models/vitstr/system.py:51-60): the constructor defaults class_token=True, global_pool='token',
        num_classes > 0 — a learned `cls_token[1, 1, E]` is prepended to the patch tokens BEFORE `pos_embed[1, N + 1, E]`
        is added (`_pos_embed` with `no_embed_class=False`), `norm` is the final LayerNorm (`fc_norm` is only used with
        average pooling) and `head = Linear(E, num_classes)`.  ViTSTR overrides `forward` and calls `forward_features`
        and `head` itself, so `forward_head` / pooling is never reached.
    """

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, global_pool='token',
                 embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, class_token=True,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, embed_layer=PatchEmbed, **_unused):
        super().__init__()
        assert (num_classes == 0 and global_pool == '' and not class_token) or (num_classes > 0 and global_pool == 'token' and class_token), \
            'stand-in covers only the PARSeq-encoder and the ViTSTR configurations'
        assert drop_rate == 0 and attn_drop_rate == 0 and drop_path_rate == 0 and qkv_bias
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if class_token else None
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + (1 if class_token else 0), embed_dim) * 0.02)
        self.blocks = nn.Sequential(*[
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)
        ])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        if self.cls_token is not None:
            nn.init.normal_(self.cls_token, std=1e-6)
        named_apply(self._init_vit, self)

    @staticmethod
    def _init_vit(module, name=''):
        if isinstance(module, nn.Linear):
            nn.init.trunc_normal_(module.weight, std=0.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def forward_features(self, x):
        x = self.patch_embed(x)
        if self.cls_token is not None:
            x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        x = x + self.pos_embed
        x = self.blocks(x)
        return self.norm(x)

    def forward(self, x):
        return self.forward_features(x)


def install():
    """Register the stand-in as `timm.models.helpers` / `timm.models.vision_transformer` in sys.modules."""
    if 'timm' in sys.modules and not getattr(sys.modules['timm'], '_parseq_amd_standin', False):
        raise RuntimeError('a real timm is importable; use it instead of the stand-in')
    timm = types.ModuleType('timm')
    timm._parseq_amd_standin = True
    models = types.ModuleType('timm.models')
    helpers = types.ModuleType('timm.models.helpers')
    vit = types.ModuleType('timm.models.vision_transformer')
    helpers.named_apply = named_apply
    vit.PatchEmbed = PatchEmbed
    vit.VisionTransformer = VisionTransformer
    timm.models = models
    models.helpers = helpers
    models.vision_transformer = vit
    sys.modules.update({
        'timm': timm, 'timm.models': models,
        'timm.models.helpers': helpers, 'timm.models.vision_transformer': vit,
    })
