"""PARSeq "system": what `hubconf.py`, `test.py`, `read.py` and `bench.py` of the reference actually touch.

Mirrors `strhub/models/parseq/system.py:33-88` (constructor arguments, `.model`, `.forward`) and the parts of
`strhub/models/base.py` the inference path uses (`BatchResult` :36-44, `_eval_step` :112-143, `test_step` :179-180,
`.tokenizer`, `.charset_adapter`, `.bos_id/.eos_id/.pad_id` :185-192).  The reference builds on
`pytorch_lightning.LightningModule`; this is a plain `nn.Module` that provides the attributes those scripts use
(`.hparams`, `.device`, `.eval()`, `.to()`), because Lightning is framework glue outside the hot path.

Row N3 (training step): the permutation sampler and the attention-mask construction
(`system.py:90-166`, host logic, bit-identical to the reference under the same numpy / torch seeds) live here;
`training_step` (`system.py:168-199`) returns a loss whose `backward()` fills `.grad` of every parameter — forward and
backward run on the device (parseq_amd/train.py); `permutation_loss` is the forward-only evaluation with the inference kernels.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from itertools import permutations
from typing import Any, Optional, Sequence

import numpy as np
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
    with _native.guard(flat):
        _native.check(_native.lib().parseq_cross_entropy(_native.ptr(flat), _native.ptr(tgt), flat.shape[0], flat.shape[1], system.pad_id,
                                                         _native.ptr(loss), _native.ptr(numel), _native.ptr(ws), _native.stream_ptr(flat)))
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


# Of the 24 orderings of four characters (lexicographic index), the twelve the reference keeps when mirrored pairs are
# requested, so that no kept ordering is the reverse of another (system.py:110-111; a fact of the algorithm, restated).
_POOL4_MIRRORED = (0, 3, 4, 6, 9, 10, 12, 16, 17, 18, 19, 21)


def sample_char_orders(num_chars: int, max_gen_perms: int, perm_forward: bool, perm_mirrored: bool,
                       rng: np.random.Generator) -> Tensor:
    """The character orderings of `gen_tgt_perms` before <bos> / <eos> are attached (system.py:96-132): [K', num_chars]
    int64 on the CPU.  Draws from `rng` (pool branch, < 5 characters) or from torch's default CPU generator
    (`torch.randperm`, >= 5 characters) in the reference's order, so equal seeds give equal orderings AS LONG AS THE
    REFERENCE ALSO DRAWS ON THE CPU: the reference calls `torch.randperm(..., device=self._device)` (system.py:128), which on a
    GPU run consumes the device generator — a different stream — so seeded GPU training of the reference is not reproduced
    permutation for permutation (the orderings are identically distributed; the goldens in tests/golden/perms.json were
    minted with the reference on the CPU).  The pool branch also departs from the reference where the reference crashes
    (`perm_forward=False` with fewer than five characters, see below)."""
    forward = [torch.arange(num_chars)] if perm_forward else []
    limit = math.factorial(num_chars) // (2 if perm_mirrored else 1)
    want = min(max_gen_perms, limit) - len(forward)
    if num_chars < 5:
        keep = _POOL4_MIRRORED if (num_chars == 4 and perm_mirrored) else range(limit)
        every = list(permutations(range(num_chars)))
        pool = torch.tensor([every[i] for i in keep][1 if perm_forward else 0:], dtype=torch.long).reshape(-1, num_chars)
        # (with perm_forward off the reference fails here — it stacks an empty list, system.py:122; this draws from the pool)
        drawn = [pool[rng.choice(len(pool), size=want, replace=False)]] if len(pool) else []
        return torch.cat(([torch.stack(forward)] if forward else []) + drawn)
    return torch.stack(forward + [torch.randperm(num_chars) for _ in range(want)])


def gen_tgt_perms(tgt: Tensor, max_gen_perms: int, perm_forward: bool, perm_mirrored: bool, rng: np.random.Generator) -> Tensor:
    """PARSeq.gen_tgt_perms (system.py:90-150): position orderings [K, T + 2] shared by the whole batch — column 0 is <bos>,
    the last column <eos>; row 1 (when present) is the right-to-left ordering with <eos> generated first."""
    num_chars = tgt.shape[1] - 2
    if num_chars == 1:
        return torch.arange(3).unsqueeze(0)
    orders = sample_char_orders(num_chars, max_gen_perms, perm_forward, perm_mirrored, rng)
    if perm_mirrored:
        orders = torch.stack([orders, orders.flip(-1)], dim=1).reshape(-1, num_chars)      # pairs next to each other
    k = len(orders)
    perms = torch.cat([orders.new_zeros(k, 1), orders + 1, orders.new_full((k, 1), num_chars + 1)], dim=1)
    if k > 1:
        perms[1, 1:] = torch.arange(num_chars + 1, 0, -1)
    return perms


def generate_attn_masks(perm: Tensor):
    """PARSeq.generate_attn_masks (system.py:152-166): bool (content_mask, query_mask), True = masked.  Position q sees
    position k only if k precedes q in `perm`; the query mask also hides q itself and drops the <bos> query row, both drop
    the <eos> key column.  Computed from each position's rank in `perm`."""
    sz = perm.shape[0]
    rank = torch.empty(sz, dtype=torch.long, device=perm.device)
    rank[perm] = torch.arange(sz, device=perm.device)
    after = rank.unsqueeze(0) > rank.unsqueeze(1)
    hidden = after | torch.eye(sz, dtype=torch.bool, device=perm.device)
    return after[:-1, :-1].contiguous(), hidden[1:, :-1].contiguous()


def permutation_loss(system, images: Tensor, labels, perms: Optional[Tensor] = None):
    """Forward half of PARSeq.training_step (system.py:168-199) with dropout off: one `encode`, one teacher-forced decode
    per permutation (the depth-1 decoder only reads the query mask), the cross-entropy of each weighted by its count of
    non-<pad> targets; <eos> targets are dropped after the first two permutations (:191-195).  Everything after the mask
    construction runs on the device (`parseq_decode_logits`, `parseq_cross_entropy`); no host synchronisation.
    Returns (loss, per-permutation losses [K], per-permutation target counts [K], perms)."""
    from . import _native
    dev = system.device
    tgt = system.tokenizer.encode(labels, dev)
    if perms is None:
        perms = system.gen_tgt_perms(tgt)
    tgt_in, tgt_out = tgt[:, :-1], tgt[:, 1:]
    L = tgt_in.shape[1]
    masks = torch.stack([generate_attn_masks(p)[1] for p in perms.cpu()]).to(torch.uint8).to(dev)       # one upload
    padding = ((tgt_in == system.pad_id) | (tgt_in == system.eos_id))
    system.model.encode(images)                      # leaves the cross-attention K / V of these images on the device
    targets = [tgt_out.to(torch.int32).contiguous().view(-1),
               torch.where(tgt_out == system.eos_id, system.pad_id, tgt_out).to(torch.int32).contiguous().view(-1)]
    K = len(perms)
    losses = torch.empty(K, dtype=torch.float32, device=dev)
    counts = torch.empty(K, dtype=torch.int32, device=dev)
    ws = torch.empty(tgt_out.numel(), dtype=torch.float32, device=dev)
    for i in range(K):
        logits = system.model.decode_logits(tgt_in, 0, L, padding, masks[i])
        flat = logits.view(-1, logits.shape[-1])
        with _native.guard(flat):
            _native.check(_native.lib().parseq_cross_entropy(_native.ptr(flat), _native.ptr(targets[min(i // 2, 1)]), flat.shape[0],
                                                             flat.shape[1], system.pad_id, _native.ptr(losses[i:]),
                                                             _native.ptr(counts[i:]), _native.ptr(ws), _native.stream_ptr(flat)))
    weights = counts.float()
    return (losses * weights).sum() / weights.sum(), losses, counts, perms


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
        # permutation sampling state (system.py:81-85)
        self.rng = np.random.default_rng()
        self.max_gen_perms = perm_num // 2 if perm_mirrored else perm_num
        self.perm_forward, self.perm_mirrored = perm_forward, perm_mirrored

    @property
    def device(self) -> torch.device:
        return self.model._device

    @property
    def precision(self) -> str:
        return self.model.precision

    @precision.setter
    def precision(self, value: str) -> None:
        self.model.precision = value

    def forward(self, images: Tensor, max_length: Optional[int] = None, slot: Optional[int] = None) -> Tensor:
        """Inference (system.py:87-88): images [N, 3, H, W] -> logits [N, L, C].  (`slot`: see model.PARSeq.forward.)"""
        return self.model.forward(self.tokenizer, images, max_length, slot)

    def forward_with_length(self, images: Tensor, max_length: Optional[int] = None, slot: Optional[int] = None):
        """(logits of all num_steps positions, L): `forward` is `logits[:, :L]`.  L < num_steps only for AR decoding without
        refinement and `max_length=None` (the batch-level early exit of model.py:144-145)."""
        return self.model.forward(self.tokenizer, images, max_length, slot, return_length=True)

    # ---- evaluation glue used by the reference's test.py:121-126 ------------------------------------------------
    def _eval_step(self, batch, validation: bool):
        return eval_step(self, batch, validation)

    def forward_logits_loss(self, images: Tensor, labels):
        return forward_logits_loss(self, images, labels)

    # ---- row N3, forward half (system.py:90-199) ------------------------------------------------------------------
    def gen_tgt_perms(self, tgt: Tensor) -> Tensor:
        """Orderings are drawn on the host (a few hundred bytes per step) whatever device `tgt` lives on."""
        return gen_tgt_perms(tgt, self.max_gen_perms, self.perm_forward, self.perm_mirrored, self.rng)

    def generate_attn_masks(self, perm: Tensor):
        return generate_attn_masks(perm)

    def training_step(self, batch, batch_idx):
        """system.py:168-199.  With autograd enabled: the loss as ONE autograd node whose backward deposits the gradient of every
        parameter (forward and backward run fused on the device in fp32, parseq_amd/train.py; dropout as `self.training` says).
        Under `torch.no_grad()`: the forward-only evaluation of the same loss with the inference kernels (`permutation_loss`)."""
        images, labels = batch
        if torch.is_grad_enabled():
            from .train import training_step_loss
            return training_step_loss(self, images, labels)
        return permutation_loss(self, images, labels)[0]

    def validation_step(self, batch, batch_idx):
        return self._eval_step(batch, True)

    def test_step(self, batch, batch_idx):
        return self._eval_step(batch, False)
