"""CPU restatement (test infrastructure) of the reference's ViTSTR inference path — SURVEY.md section 8f row N4.

  strhub/models/vitstr/system.py:76-82   forward(images, max_length): seqlen = max_length + 2 ([GO] and [s]);
                                         logits = model.forward(images, seqlen)[:, 1:]  (the class-token position is dropped)
  strhub/models/vitstr/model.py:20-28    ViTSTR(VisionTransformer).forward(x, seqlen): forward_features(x)[:, :seqlen] -> head
  timm VisionTransformer (un-vendored, restated in oracle/timm_standin.py and oracle/parseq_oracle.vit_features):
                                         class token prepended, pos_embed[1, N + 1, E], 12 pre-LN blocks, final LayerNorm

The encoder arithmetic is the same function the PARSeq oracle uses (`parseq_oracle.vit_features`), so the rounding-aware
bf16 mode is available here too.  Pinned by tests/golden/vitstr.* (oracle/make_golden_vitstr.py runs the reference's
own vitstr/model.py on the timm stand-in).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

from torch import Tensor

from . import parseq_oracle as O
from .synth import synth_state_dict as _synth


def vitstr_config(embed_dim: int = 384, num_heads: int = 6, img_size=(32, 128), patch_size=(4, 8)) -> O.OracleConfig:
    """configs/model/vitstr.yaml (embed_dim 384, 6 heads) + configs/experiment/vitstr.yaml:5-7 (32x128 crops, 4x8 patches);
    depth 12, mlp_ratio 4 are fixed in vitstr/system.py:54-56."""
    return O.OracleConfig(img_size=tuple(img_size), patch_size=tuple(patch_size), embed_dim=embed_dim, enc_num_heads=num_heads,
                          enc_mlp_ratio=4, enc_depth=12)


def state_dict_spec(cfg: O.OracleConfig) -> 'OrderedDict[str, tuple]':
    """Parameter names / shapes of the reference's `ViTSTR(VisionTransformer)` (timm key layout, class token + head)."""
    E, N = cfg.embed_dim, cfg.num_patches
    ph, pw = cfg.patch_size
    spec = OrderedDict()
    spec['cls_token'] = (1, 1, E)
    spec['pos_embed'] = (1, N + 1, E)
    spec['patch_embed.proj.weight'] = (E, 3, ph, pw)
    spec['patch_embed.proj.bias'] = (E,)
    for i in range(cfg.enc_depth):
        p = f'blocks.{i}.'
        spec[p + 'norm1.weight'] = (E,)
        spec[p + 'norm1.bias'] = (E,)
        spec[p + 'attn.qkv.weight'] = (3 * E, E)
        spec[p + 'attn.qkv.bias'] = (3 * E,)
        spec[p + 'attn.proj.weight'] = (E, E)
        spec[p + 'attn.proj.bias'] = (E,)
        spec[p + 'norm2.weight'] = (E,)
        spec[p + 'norm2.bias'] = (E,)
        spec[p + 'mlp.fc1.weight'] = (E * cfg.enc_mlp_ratio, E)
        spec[p + 'mlp.fc1.bias'] = (E * cfg.enc_mlp_ratio,)
        spec[p + 'mlp.fc2.weight'] = (E, E * cfg.enc_mlp_ratio)
        spec[p + 'mlp.fc2.bias'] = (E,)
    spec['norm.weight'] = (E,)
    spec['norm.bias'] = (E,)
    spec['head.weight'] = (cfg.num_tokens - 2, E)       # "We don't predict <bos> nor <pad>" (vitstr/system.py:50,59)
    spec['head.bias'] = (cfg.num_tokens - 2,)
    return spec


def synth_state_dict(cfg: O.OracleConfig, seed: int = 0):
    """Per-key seeded 'trained-like' weights (oracle/synth.py) for the ViTSTR parameter set."""
    return _synth(cfg, seed, spec=state_dict_spec(cfg))


def forward(sd: dict, cfg: O.OracleConfig, images: Tensor, max_length: Optional[int] = None, rounding: Optional[str] = None) -> Tensor:
    """Returns logits [B, min(max_length, max_label_length) + 1, num_tokens - 2]."""
    max_length = cfg.max_label_length if max_length is None else min(max_length, cfg.max_label_length)
    seqlen = max_length + 2
    x = O.vit_features(sd, '', cfg, images, rounding)[:, :seqlen]
    logits = O._linear(x, sd['head.weight'], sd['head.bias'], rounding)
    return logits[:, 1:]
