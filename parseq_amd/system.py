"""PARSeq "system": what `hubconf.py`, `test.py`, `read.py` and `bench.py` of the reference actually touch.

Mirrors `strhub/models/parseq/system.py:33-88` (constructor arguments, `.model`, `.forward`) and the parts of
`strhub/models/base.py` the inference path uses (`BatchResult` :36-44, `_eval_step` :112-143, `test_step` :179-180,
`.tokenizer`, `.charset_adapter`, `.bos_id/.eos_id/.pad_id` :185-192).  The reference builds on
`pytorch_lightning.LightningModule`; this is a plain `nn.Module` that provides the attributes those scripts use
(`.hparams`, `.device`, `.eval()`, `.to()`), because Lightning is framework glue outside the hot path.
Training (`training_step`, permutation sampling, optimiser) is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from .model import PARSeq as Model
from .tokenizer import CharsetAdapter, Tokenizer


class AttributeDict(dict):
    """dict with attribute access — stands in for Lightning's `hparams` container (`save_hyperparameters`)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value


@dataclass
class BatchResult:
    num_samples: int
    correct: int
    ned: float
    confidence: float
    label_length: int
    loss: Optional[Tensor]
    loss_numel: Optional[int]


def edit_distance(a: str, b: str) -> int:
    """Levenshtein distance (unit costs) — what `nltk.edit_distance` computes with its defaults (base.py:20,137)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def forward_logits_loss(system, images: Tensor, labels):
    """CrossEntropySystem.forward_logits_loss (strhub/models/base.py:194-201): logits for max_len = longest label, the mean
    cross-entropy over non-<pad> targets and their count, the loss computed on the device (`parseq_cross_entropy`)."""
    from . import _native
    targets = system.tokenizer.encode(labels, images.device)[:, 1:]           # discard <bos>
    max_len = targets.shape[1] - 1                                             # exclude <eos> from the count
    logits = system.forward(images, max_len)
    if logits.shape[1] != targets.shape[1]:
        raise RuntimeError(f'labels longer than max_label_length: targets {targets.shape[1]} positions, logits {logits.shape[1]}')
    flat = logits.float().contiguous().view(-1, logits.shape[-1])
    tgt = targets.to(torch.int32).contiguous().view(-1)
    loss = torch.empty((), dtype=torch.float32, device=images.device)
    numel = torch.empty((), dtype=torch.int32, device=images.device)
    ws = torch.empty(flat.shape[0], dtype=torch.float32, device=images.device)
    _native.check(_native.lib().parseq_cross_entropy(_native.ptr(flat), _native.ptr(tgt), flat.shape[0], flat.shape[1], system.pad_id,
                                                     _native.ptr(loss), _native.ptr(numel), _native.ptr(ws), _native.stream_ptr()))
    return logits, loss, numel


def eval_step(system, batch, validation: bool):
    """BaseSystem._eval_step (strhub/models/base.py:112-143) for any system with `.forward`, `.tokenizer`, `.charset_adapter`."""
    images, labels = batch
    with torch.inference_mode():
        if validation:
            logits, loss, loss_numel = forward_logits_loss(system, images, labels)
        else:
            # at test time no max_length is given (base.py:123-130): the test charset may shorten the labels
            logits, loss, loss_numel = system.forward(images), None, None
        # base.py:132-137 on the device: soft-max, greedy pick, first-EOS cut and prob.prod() in one kernel (row N1)
        preds, confs = system.tokenizer.read(logits)
    correct = total = label_length = 0
    ned = confidence = 0.0
    for pred, conf, gt in zip(preds, confs.tolist(), labels):
        confidence += conf
        pred = system.charset_adapter(pred)
        ned += edit_distance(pred, gt) / max(len(pred), len(gt), 1)
        correct += int(pred == gt)
        total += 1
        label_length += len(pred)
    return dict(output=BatchResult(total, correct, ned, confidence, label_length, loss, loss_numel))


class PARSeq(nn.Module):

    def __init__(self, charset_train: str, charset_test: str, max_label_length: int, batch_size: int, lr: float,
                 warmup_pct: float, weight_decay: float, img_size: Sequence[int], patch_size: Sequence[int],
                 embed_dim: int, enc_num_heads: int, enc_mlp_ratio: int, enc_depth: int, dec_num_heads: int,
                 dec_mlp_ratio: int, dec_depth: int, perm_num: int, perm_forward: bool, perm_mirrored: bool,
                 decode_ar: bool, refine_iters: int, dropout: float, **kwargs: Any) -> None:
        super().__init__()
        precision = kwargs.pop('precision', None)
        hp = dict(charset_train=charset_train, charset_test=charset_test, max_label_length=max_label_length,
                  batch_size=batch_size, lr=lr, warmup_pct=warmup_pct, weight_decay=weight_decay, img_size=list(img_size),
                  patch_size=list(patch_size), embed_dim=embed_dim, enc_num_heads=enc_num_heads,
                  enc_mlp_ratio=enc_mlp_ratio, enc_depth=enc_depth, dec_num_heads=dec_num_heads,
                  dec_mlp_ratio=dec_mlp_ratio, dec_depth=dec_depth, perm_num=perm_num, perm_forward=perm_forward,
                  perm_mirrored=perm_mirrored, decode_ar=decode_ar, refine_iters=refine_iters, dropout=dropout)
        hp.update(kwargs)            # name / _target_ / _convert_ ... are swallowed exactly as the reference's **kwargs does
        self.hparams = AttributeDict(hp)
        self.tokenizer = Tokenizer(charset_train)
        self.charset_adapter = CharsetAdapter(charset_test)
        self.bos_id, self.eos_id, self.pad_id = self.tokenizer.bos_id, self.tokenizer.eos_id, self.tokenizer.pad_id
        self.batch_size, self.lr, self.warmup_pct, self.weight_decay = batch_size, lr, warmup_pct, weight_decay
        self.model = Model(len(self.tokenizer), max_label_length, img_size, patch_size, embed_dim, enc_num_heads,
                           enc_mlp_ratio, enc_depth, dec_num_heads, dec_mlp_ratio, dec_depth, decode_ar, refine_iters,
                           dropout, precision=precision)

    @property
    def device(self) -> torch.device:
        return self.model._device

    @property
    def precision(self) -> str:
        return self.model.precision

    @precision.setter
    def precision(self, value: str) -> None:
        self.model.precision = value

    def forward(self, images: Tensor, max_length: Optional[int] = None, slot: int = 0) -> Tensor:
        """Inference (system.py:87-88): images [N, 3, H, W] -> logits [N, L, C].  (`slot`: see model.PARSeq.forward.)"""
        return self.model.forward(self.tokenizer, images, max_length, slot)

    # ---- evaluation glue used by the reference's test.py:121-126 ------------------------------------------------
    def _eval_step(self, batch, validation: bool):
        return eval_step(self, batch, validation)

    def forward_logits_loss(self, images: Tensor, labels):
        return forward_logits_loss(self, images, labels)

    def validation_step(self, batch, batch_idx):
        return self._eval_step(batch, True)

    def test_step(self, batch, batch_idx):
        return self._eval_step(batch, False)
