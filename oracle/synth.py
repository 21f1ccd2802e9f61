"""Deterministic synthetic weights and inputs for parity tests.   *** TEST INFRASTRUCTURE ***

No pretrained PARSeq weights are reachable (no network), and the reference initialiser zeroes every bias and
sets every LayerNorm to (1, 0) (strhub/models/utils.py:107-125, timm ViT init) — a kernel that dropped a bias or
an LN affine term would still pass parity on such weights.  So tests use a synthetic `state_dict` with
non-trivial values everywhere, generated per key from a seed so that the oracle (CPU) and the HIP path load
bit-identical tensors on any box without shipping 95 MB files.

The key set / shapes are those of the reference inner model (`strhub/models/parseq/model.py:56-67`,
`modules.py:33-43`, timm ViT), listed in SURVEY.md section 8(b); `oracle/make_golden.py` verifies them by
`load_state_dict(strict=True)` into the reference's own module.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import torch

from .parseq_oracle import OracleConfig


def state_dict_spec(cfg: OracleConfig) -> 'OrderedDict[str, tuple]':
    E = cfg.embed_dim
    N = cfg.num_patches
    ph, pw = cfg.patch_size
    spec = OrderedDict()
    spec['pos_queries'] = (1, cfg.max_label_length + 1, E)
    spec['encoder.pos_embed'] = (1, N, E)
    spec['encoder.patch_embed.proj.weight'] = (E, 3, ph, pw)
    spec['encoder.patch_embed.proj.bias'] = (E,)
    for i in range(cfg.enc_depth):
        p = f'encoder.blocks.{i}.'
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
    spec['encoder.norm.weight'] = (E,)
    spec['encoder.norm.bias'] = (E,)
    for i in range(cfg.dec_depth):
        p = f'decoder.layers.{i}.'
        for attn in ('self_attn', 'cross_attn'):
            spec[p + attn + '.in_proj_weight'] = (3 * E, E)
            spec[p + attn + '.in_proj_bias'] = (3 * E,)
            spec[p + attn + '.out_proj.weight'] = (E, E)
            spec[p + attn + '.out_proj.bias'] = (E,)
        F_ = E * cfg.dec_mlp_ratio
        spec[p + 'linear1.weight'] = (F_, E)
        spec[p + 'linear1.bias'] = (F_,)
        spec[p + 'linear2.weight'] = (E, F_)
        spec[p + 'linear2.bias'] = (E,)
        for n in ('norm1', 'norm2', 'norm_q', 'norm_c'):
            spec[p + n + '.weight'] = (E,)
            spec[p + n + '.bias'] = (E,)
    spec['decoder.norm.weight'] = (E,)
    spec['decoder.norm.bias'] = (E,)
    spec['head.weight'] = (cfg.num_tokens - 2, E)
    spec['head.bias'] = (cfg.num_tokens - 2,)
    spec['text_embed.embedding.weight'] = (cfg.num_tokens, E)
    return spec


def _key_seed(seed: int, key: str) -> int:
    h = hashlib.sha256(f'{seed}:{key}'.encode()).digest()
    return int.from_bytes(h[:7], 'little')


def synth_state_dict(cfg: OracleConfig, seed: int = 0, eos_bias: float = 2.5, gain: float = 1.0, spec=None) -> 'OrderedDict[str, torch.Tensor]':
    """Per-key seeded weights in a 'trained-like' regime: Linear/Conv weights ~ N(0, gain/sqrt(fan_in)) so that
    activations stay O(1) and attention is far from uniform; biases ~ N(0, 0.1); LayerNorm weight ~ 1 + N(0, 0.1),
    bias ~ N(0, 0.1); embeddings / positional tables ~ N(0, 0.5 or 0.05).  `eos_bias` lifts the [E] logit so that
    end-of-sequence actually occurs at mixed positions (exercises early exit and the refinement padding mask).
    CPU generator => identical bits on every x86 box with the same torch build."""
    sd = OrderedDict()
    for key, shape in (spec if spec is not None else state_dict_spec(cfg)).items():
        g = torch.Generator(device='cpu').manual_seed(_key_seed(seed, key))
        n = torch.randn(shape, generator=g, dtype=torch.float32)
        leaf = key.rsplit('.', 1)[-1]
        is_norm = any(t in key for t in ('norm1', 'norm2', 'norm_q', 'norm_c', '.norm.', 'encoder.norm', 'decoder.norm')) or key.startswith('norm.')
        if key == 'pos_queries':
            t = 0.5 * n
        elif key in ('encoder.pos_embed', 'pos_embed', 'cls_token'):
            t = 0.3 * n
        elif key == 'text_embed.embedding.weight':
            t = 0.05 * n          # multiplied by sqrt(E) ~ 19.6 in TokenEmbedding
        elif is_norm:
            t = 1.0 + 0.1 * n if leaf == 'weight' else 0.1 * n
        elif leaf in ('weight', 'in_proj_weight'):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = (gain / fan_in ** 0.5) * n
            if 'attn.qkv' in key or 'in_proj' in key:
                t = 1.6 * t       # sharper attention logits
        else:                     # biases
            t = 0.1 * n
        sd[key] = t.contiguous()
    sd['head.bias'][cfg.eos_id] += eos_bias
    return sd


def synth_images(batch: int, cfg: OracleConfig, seed: int = 1234) -> torch.Tensor:
    """Crops as the reference's transform produces them: float32 in [-1, 1] (strhub/data/module.py:78-80,
    Normalize(0.5, 0.5) after ToTensor).  SURVEY.md section 8(d)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.rand(batch, 3, cfg.img_size[0], cfg.img_size[1], generator=g, dtype=torch.float32) * 2 - 1


def state_dict_fingerprint(sd) -> float:
    """Cheap order-dependent checksum used by goldens to detect RNG drift between boxes."""
    acc = 0.0
    for i, (k, v) in enumerate(sd.items()):
        acc += (i + 1) * float(v.double().abs().sum())
    return acc


CHARSET_36 = "0123456789abcdefghijklmnopqrstuvwxyz"                       # configs/charset/36_lowercase.yaml:3 (a fact, restated)
CHARSET_62 = CHARSET_36 + "ABCDEFGHIJKLMNOPQRSTUVWXYZ"                      # configs/charset/62_mixed-case.yaml:3
CHARSET_200 = CHARSET_62 + "".join(chr(0x4E00 + i) for i in range(138))      # 62 + 138 CJK ideographs: a charset the reference accepts like any other string


def charset_config(base: OracleConfig, charset: str, **overrides) -> OracleConfig:
    """The model shape `Tokenizer(charset)` implies (strhub/data/utils.py:101-111: [E] = 0, the characters, [B], [P])."""
    import dataclasses
    n = len(charset)
    return dataclasses.replace(base, num_tokens=n + 3, bos_id=n + 1, pad_id=n + 2, **overrides)


CONFIGS = {
    # configs/model/parseq.yaml:5-14 + configs/main.yaml:9-10 + configs/charset/94_full.yaml (94 chars -> 97 tokens)
    'parseq': OracleConfig(),
    # configs/experiment/parseq-tiny.yaml:5-9
    'parseq-tiny': OracleConfig(embed_dim=192, enc_num_heads=3, dec_num_heads=6),
    # configs/experiment/parseq-patch16-224.yaml:5-7 (row N4): 14 x 14 = 196 visual tokens, 768-wide patches
    'parseq-patch16-224': OracleConfig(img_size=(224, 224), patch_size=(16, 16)),
}

# Configurations reached through the hub keyword arguments the reference accepts (strhub/models/utils.py:41 `config.update(kwargs)`):
# name -> (experiment, keyword overrides, EOS bias of the synthetic head: narrower heads need less of it for [E] to land at mixed
# positions).  Goldens: oracle/make_golden_hub.py -> tests/golden/<name>.*; the name is NOT an experiment — build with
# create_model(experiment, **overrides) and load variant_state_dict(name).
HUB_VARIANTS = {
    'parseq_c36_len10': ('parseq', {'charset_train': CHARSET_36, 'max_label_length': 10}, 0.5),
    'parseq_c62': ('parseq', {'charset_train': CHARSET_62}, 1.25),
    'parseq-tiny_c62_len10': ('parseq-tiny', {'charset_train': CHARSET_62, 'max_label_length': 10}, 1.5),
    # more classes than one 128-column head tile of the fused AR step (decoder_step.h): the AR loop must take the per-operation kernels; 201-wide
    # head rows are not 16-byte multiples; non-ASCII characters through the tokenizer
    'parseq_c200': ('parseq', {'charset_train': CHARSET_200}, 1.5),
}


def variant_config(name: str) -> OracleConfig:
    experiment, kw, _ = HUB_VARIANTS[name]
    extra = {'max_label_length': kw['max_label_length']} if 'max_label_length' in kw else {}
    return charset_config(CONFIGS[experiment], kw['charset_train'], **extra)


def variant_state_dict(name: str, seed: int = 0):
    return synth_state_dict(variant_config(name), seed, eos_bias=HUB_VARIANTS[name][2])


# ViTSTR through the same hub keyword arguments (strhub/models/vitstr/system.py:41-60: the head is len(charset) + 1 wide, forward returns
# max_label_length + 1 positions): name -> keyword overrides of create_model('vitstr', ...).  Goldens: oracle/make_golden_hub.py.
VITSTR_HUB_VARIANTS = {
    'vitstr_c36_len10': {'charset_train': CHARSET_36, 'max_label_length': 10},
}


def vitstr_variant_config(name: str) -> OracleConfig:
    from .vitstr_oracle import vitstr_config
    kw = VITSTR_HUB_VARIANTS[name]
    extra = {'max_label_length': kw['max_label_length']} if 'max_label_length' in kw else {}
    return charset_config(vitstr_config(), kw['charset_train'], **extra)
