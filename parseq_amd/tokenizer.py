"""Charset / token-id handling on the edge of the hot path.

Mirrors the interface of the reference's `strhub/data/utils.py` (`CharsetAdapter` :26-43, `BaseTokenizer` :46-99,
`Tokenizer` :102-129): same class and method names, same id assignment ([E] = 0, characters 1..len(charset),
[B], [P] last), same greedy decode + truncate-at-first-EOS rule — string parity with the reference is judged on the
output of `Tokenizer.decode`.  `CharsetAdapter`, `BaseTokenizer._tok2ids / _ids2tok / __len__` and `Tokenizer._filter` are
near-verbatim behavioural mirrors of the reference's few lines each (a names-and-semantics contract leaves no other way to
write them); `encode`, `decode`, `decode_logits` and `read` are this repository's own.  `decode` is the reference's host-side routine;
`decode_logits` / `read` give the same result from raw logits with the numeric part (soft-max, greedy pick, first-EOS cut,
confidence product) done by the HIP post-processing kernel (`parseq_postprocess`).
"""
from __future__ import annotations

import re
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor


class CharsetAdapter:
    """Maps a label into the target charset: case-fold if the charset is single-case, then drop foreign characters."""

    def __init__(self, target_charset: str) -> None:
        self.lowercase_only = target_charset == target_charset.lower()
        self.uppercase_only = target_charset == target_charset.upper()
        self.unsupported = re.compile(f'[^{re.escape(target_charset)}]')

    def __call__(self, label: str) -> str:
        if self.lowercase_only:
            label = label.lower()
        elif self.uppercase_only:
            label = label.upper()
        return self.unsupported.sub('', label)


class BaseTokenizer:
    def __init__(self, charset: str, specials_first: Tuple[str, ...] = (), specials_last: Tuple[str, ...] = ()) -> None:
        self._itos = tuple(specials_first) + tuple(charset) + tuple(specials_last)
        self._stoi = {s: i for i, s in enumerate(self._itos)}

    def __len__(self) -> int:
        return len(self._itos)

    def _tok2ids(self, tokens: str) -> List[int]:
        return [self._stoi[s] for s in tokens]

    def _ids2tok(self, token_ids: Sequence[int], join: bool = True):
        tokens = [self._itos[i] for i in token_ids]
        return ''.join(tokens) if join else tokens

    def encode(self, labels: List[str], device: Optional[torch.device] = None) -> Tensor:
        raise NotImplementedError

    def _filter(self, probs: Tensor, ids: Tensor):
        raise NotImplementedError

    def decode(self, token_dists: Tensor, raw: bool = False) -> Tuple[list, List[Tensor]]:
        """token_dists: probabilities [N, L, C].  Returns (labels, per-label probability tensors)."""
        # one device->host transfer for the whole batch instead of one per row
        probs_all, ids_all = token_dists.max(-1)
        ids_all = ids_all.cpu()          # probabilities stay on the caller's device, as in the reference
        batch_tokens, batch_probs = [], []
        for probs, ids in zip(probs_all, ids_all):
            if not raw:
                probs, ids = self._filter(probs, ids)
            else:
                ids = ids.tolist()
            batch_tokens.append(self._ids2tok(ids, not raw))
            batch_probs.append(probs)
        return batch_tokens, batch_probs


class Tokenizer(BaseTokenizer):
    BOS = '[B]'
    EOS = '[E]'
    PAD = '[P]'

    def __init__(self, charset: str) -> None:
        super().__init__(charset, (self.EOS,), (self.BOS, self.PAD))
        self.eos_id, self.bos_id, self.pad_id = (self._stoi[s] for s in (self.EOS, self.BOS, self.PAD))

    def encode(self, labels: List[str], device: Optional[torch.device] = None) -> Tensor:
        # strhub/data/utils.py:113-116 ([B] + ids + [E] per label, pad_sequence with [P]) — assembled in one numpy array: the per-row tensor
        # writes cost 3.2 ms for a batch of 384 labels, as long as the encoder's whole training forward on the device
        import numpy as np
        stoi = self._stoi
        lens = np.fromiter((len(y) for y in labels), dtype=np.int64, count=len(labels))
        flat = np.fromiter((stoi[c] for y in labels for c in y), dtype=np.int64, count=int(lens.sum()))
        width = (int(lens.max()) if len(labels) else 0) + 2
        out = np.full((len(labels), width), self.pad_id, dtype=np.int64)
        out[:, 0] = self.bos_id
        rows = np.repeat(np.arange(len(labels)), lens)
        cols = np.arange(flat.size) - np.repeat(np.cumsum(lens) - lens, lens) + 1
        out[rows, cols] = flat
        out[np.arange(len(labels)), lens + 1] = self.eos_id
        out = torch.from_numpy(out)
        return out.to(device) if device is not None else out

    def _filter(self, probs: Tensor, ids: Tensor):
        ids = ids.tolist()
        try:
            eos_idx = ids.index(self.eos_id)
        except ValueError:
            eos_idx = len(ids)
        # characters stop before the first EOS; the probability list keeps the EOS probability itself
        return probs[:eos_idx + 1], ids[:eos_idx]

    # ---- device-side post-processing (row N1) ---------------------------------------------------------------------
    def _postprocess(self, logits: Tensor):
        """logits: CUDA fp32 [N, L, C].  Returns device tensors (ids int32 [N, L], lengths int32 [N], probs [N, L], conf [N])."""
        from . import _native
        if not logits.is_cuda:
            raise RuntimeError('decode_logits / read run on the GPU (no CPU fallback); use decode(logits.softmax(-1)) on the host')
        logits = logits.float().contiguous()
        n, length, classes = logits.shape
        dev = logits.device
        ids = torch.empty((n, length), dtype=torch.int32, device=dev)
        lengths = torch.empty((n,), dtype=torch.int32, device=dev)
        probs = torch.empty((n, length), dtype=torch.float32, device=dev)
        conf = torch.empty((n,), dtype=torch.float32, device=dev)
        if n:
            with _native.guard(dev):
                _native.check(_native.lib().parseq_postprocess(_native.ptr(logits), n, length, classes, self.eos_id,
                                                               _native.ptr(ids), _native.ptr(lengths), _native.ptr(probs), _native.ptr(conf),
                                                               _native.stream_ptr(dev)))
        return ids, lengths, probs, conf

    def decode_logits(self, logits: Tensor) -> Tuple[List[str], List[Tensor]]:
        """Drop-in for `decode(logits.softmax(-1))`: same labels, same per-label probability tensors (views of one device
        tensor), with one small device->host copy (ids + lengths) instead of a [N, L, C] soft-max and a .tolist() per row."""
        ids, lengths, probs, _ = self._postprocess(logits)
        ids_h, len_h = ids.cpu().numpy(), lengths.cpu().tolist()
        width = ids_h.shape[1] if ids_h.ndim == 2 else 0
        labels = [self._ids2tok(row[:k].tolist()) for row, k in zip(ids_h, len_h)]
        return labels, [probs[i, :min(k + 1, width)] for i, k in enumerate(len_h)]

    def read(self, logits: Tensor) -> Tuple[List[str], Tensor]:
        """(labels, confidence [N] on the host): what `read.py` / `_eval_step` consume (`prob.prod()` per label)."""
        ids, lengths, _, conf = self._postprocess(logits)
        ids_h, len_h = ids.cpu().numpy(), lengths.cpu().tolist()
        return [self._ids2tok(row[:k].tolist()) for row, k in zip(ids_h, len_h)], conf.cpu()
