"""Charset / token-id handling on the edge of the hot path.

Mirrors the interface of the reference's `strhub/data/utils.py` (`CharsetAdapter` :26-43, `BaseTokenizer` :46-99,
`Tokenizer` :102-129): same class and method names, same id assignment ([E] = 0, characters 1..len(charset),
[B], [P] last), same greedy decode + truncate-at-first-EOS rule — string parity with the reference is judged on the
output of `Tokenizer.decode`.  Written from scratch; CPU-side post-processing (a device-side version is row N1 of
SURVEY.md section 8f).
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
        rows = [[self.bos_id] + self._tok2ids(y) + [self.eos_id] for y in labels]
        width = max(len(r) for r in rows)
        out = torch.full((len(rows), width), self.pad_id, dtype=torch.long)
        for i, r in enumerate(rows):
            out[i, :len(r)] = torch.as_tensor(r, dtype=torch.long)
        return out.to(device) if device is not None else out

    def _filter(self, probs: Tensor, ids: Tensor):
        ids = ids.tolist()
        try:
            eos_idx = ids.index(self.eos_id)
        except ValueError:
            eos_idx = len(ids)
        # characters stop before the first EOS; the probability list keeps the EOS probability itself
        return probs[:eos_idx + 1], ids[:eos_idx]
