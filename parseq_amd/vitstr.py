"""ViTSTR on the same library.

Mirrors `strhub/models/vitstr/model.py:14-28` (class `ViTSTR(VisionTransformer)`: timm's parameter layout with a class
token and a `head`, `forward(x, seqlen)` = head of the first `seqlen` encoder tokens) and `strhub/models/vitstr/system.py:33-82`
(constructor arguments, `.model`, `forward(images, max_length)` dropping the class-token position).  As for PARSeq the
modules only hold parameters under the reference's state_dict keys; the arithmetic is `parseq_vitstr_forward` in
libparseq_hip (the PARSeq encoder kernels + class-token assembly + a plain head GEMM).
"""
from __future__ import annotations

import os
from typing import Any, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from . import _native
from .model import _Block, _NativeBacked, _NativeState, _PatchEmbed, init_weights
from .system import AttributeDict, eval_step, forward_logits_loss
from .tokenizer import CharsetAdapter, Tokenizer


class Model(_NativeBacked):
    """Parameter layout of timm's VisionTransformer with the defaults ViTSTR relies on: class token, `pos_embed` over
    N + 1 positions, final `norm`, `head = Linear(E, num_classes)`."""

    def __init__(self, img_size: Sequence[int], patch_size: Sequence[int], depth: int, mlp_ratio: int, qkv_bias: bool,
                 embed_dim: int, num_heads: int, num_classes: int, precision: Optional[str] = None) -> None:
        super().__init__()
        if not qkv_bias:
            raise ValueError('qkv_bias=False is not supported (the reference always passes True)')
        self.num_classes = num_classes
        self.precision = precision or os.environ.get('PARSEQ_AMD_PRECISION', 'bf16x3')      # the mode that meets the reference within 1e-3; 'bf16' = throughput mode
        self._cfg = dict(img_size=tuple(img_size), patch_size=tuple(patch_size), embed_dim=embed_dim, depth=depth,
                         num_heads=num_heads, mlp_ratio=mlp_ratio)
        n_tok = (img_size[0] // patch_size[0]) * (img_size[1] // patch_size[1])
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.empty(1, n_tok + 1, embed_dim))
        self.patch_embed = _PatchEmbed(patch_size, embed_dim)
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():      # timm ViT init: trunc-normal(0.02) Linear weights, zero biases
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
        object.__setattr__(self, '_native_state', _NativeState())
        self.max_label_length = 25       # set by the system; bounds the head slice the library will compute

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def _make_native_config(self):
        c = self._cfg
        n = self.num_classes + 2
        return _native.ParseqConfig(
            img_h=c['img_size'][0], img_w=c['img_size'][1], patch_h=c['patch_size'][0], patch_w=c['patch_size'][1],
            embed_dim=c['embed_dim'], enc_depth=c['depth'], enc_heads=c['num_heads'], enc_mlp_ratio=c['mlp_ratio'],
            dec_depth=0, dec_heads=0, dec_mlp_ratio=0, num_tokens=n, max_label_length=self.max_label_length,
            bos_id=n - 2, eos_id=0, pad_id=n - 1, enc_ln_eps=1e-6, dec_ln_eps=1e-5, arch=_native.ARCH_VITSTR)

    def forward(self, x: Tensor, seqlen: int = 25, slot: int = 0) -> Tensor:
        """vitstr/model.py:20-28: logits of the first `seqlen` encoder tokens (token 0 is the class token) [B, seqlen, C].

        The library returns tokens [1, seqlen) — everything the system's `forward` keeps; the class-token position, which the
        reference computes and immediately drops (system.py:81), is returned as zeros to keep the shape contract."""
        x = self._check_images(x)
        B = x.shape[0]
        if not 2 <= seqlen <= self.max_label_length + 2:
            raise RuntimeError(f'seqlen {seqlen} outside [2, {self.max_label_length + 2}]')
        plan = self._plan(B, slot)
        out = torch.zeros(B, seqlen, self.num_classes, dtype=torch.float32, device=x.device)
        body = torch.empty(B, seqlen - 1, self.num_classes, dtype=torch.float32, device=x.device)
        _native.check(_native.lib().parseq_vitstr_forward(plan, _native.ptr(x), _native.dtype_code(x.dtype), B, seqlen - 1,
                                                          _native.ptr(body), _native.stream_ptr(x)))
        out[:, 1:] = body
        return out

    def forward_sliced(self, x: Tensor, num_steps: int, slot: int = 0) -> Tensor:
        """The system-level result directly: logits of tokens [1, num_steps] -> [B, num_steps, C] (no class-token row)."""
        x = self._check_images(x)
        B = x.shape[0]
        plan = self._plan(B, slot)
        out = torch.empty(B, num_steps, self.num_classes, dtype=torch.float32, device=x.device)
        _native.check(_native.lib().parseq_vitstr_forward(plan, _native.ptr(x), _native.dtype_code(x.dtype), B, num_steps,
                                                          _native.ptr(out), _native.stream_ptr(x)))
        return out


class ViTSTR(nn.Module):
    """strhub/models/vitstr/system.py:33-82 without the Lightning training glue (out of scope, row N3)."""

    def __init__(self, charset_train: str, charset_test: str, max_label_length: int, batch_size: int, lr: float,
                 warmup_pct: float, weight_decay: float, img_size: Sequence[int], patch_size: Sequence[int], embed_dim: int,
                 num_heads: int, **kwargs: Any) -> None:
        super().__init__()
        precision = kwargs.pop('precision', None)
        hp = dict(charset_train=charset_train, charset_test=charset_test, max_label_length=max_label_length, batch_size=batch_size,
                  lr=lr, warmup_pct=warmup_pct, weight_decay=weight_decay, img_size=list(img_size), patch_size=list(patch_size),
                  embed_dim=embed_dim, num_heads=num_heads)
        hp.update(kwargs)
        self.hparams = AttributeDict(hp)
        self.tokenizer = Tokenizer(charset_train)
        self.charset_adapter = CharsetAdapter(charset_test)
        self.bos_id, self.eos_id, self.pad_id = self.tokenizer.bos_id, self.tokenizer.eos_id, self.tokenizer.pad_id
        self.batch_size, self.lr, self.warmup_pct, self.weight_decay = batch_size, lr, warmup_pct, weight_decay
        self.max_label_length = max_label_length
        # "We don't predict <bos> nor <pad>" (system.py:50): num_classes = len(tokenizer) - 2; depth 12, mlp_ratio 4 (system.py:54-56)
        self.model = Model(img_size=img_size, patch_size=patch_size, depth=12, mlp_ratio=4, qkv_bias=True, embed_dim=embed_dim,
                           num_heads=num_heads, num_classes=len(self.tokenizer) - 2, precision=precision)
        self.model.max_label_length = max_label_length
        self.model.head.apply(init_weights)

    @property
    def device(self) -> torch.device:
        return self.model._device

    @property
    def precision(self) -> str:
        return self.model.precision

    @precision.setter
    def precision(self, value: str) -> None:
        self.model.precision = value

    def no_weight_decay(self):
        return {'model.' + n for n in self.model.no_weight_decay()}

    def forward(self, images: Tensor, max_length: Optional[int] = None, slot: int = 0) -> Tensor:
        """system.py:76-82: logits [N, min(max_length, max_label_length) + 1, C]."""
        max_length = self.max_label_length if max_length is None else min(max_length, self.max_label_length)
        return self.model.forward_sliced(images, max_length + 1, slot)

    def _eval_step(self, batch, validation: bool):
        return eval_step(self, batch, validation)

    def forward_logits_loss(self, images: Tensor, labels):
        return forward_logits_loss(self, images, labels)

    def validation_step(self, batch, batch_idx):
        return self._eval_step(batch, True)

    def test_step(self, batch, batch_idx):
        return self._eval_step(batch, False)
