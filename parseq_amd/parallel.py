"""Data-parallel inference across the GPUs of one node: one process per GPU, contiguous batch shards, weights
replicated, and ONE collective per forward — an all-gather of the logits over RCCL/xGMI (BASELINE.json north_star;
SURVEY.md section 8e).  The reference has no inference-time collective; images are independent through the whole path,
so nothing else needs to cross ranks.

`torch.distributed` backend 'nccl' is RCCL on ROCm; the same code runs on 'gloo' for the CPU tests.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous split of n items into `world` shards whose sizes differ by at most one."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_logits(logits: Tensor, group=None, uniform: bool = False) -> Tensor:
    """[b_local, L, C] on every rank -> [sum b_local, L, C] on every rank (rank order).

    uniform=True: the caller guarantees identical shapes on every rank (fixed per-GPU batch, forced steps or
    refine_iters >= 1) — ONE all-gather and no host synchronisation, which is what a throughput loop wants.  Otherwise
    the shapes are exchanged first (a second small collective and a device->host read).

    Shards may differ in batch size (ragged last shard) and, with refine_iters == 0 and early exit, in L: each shard can
    stop earlier than the whole batch would.  The single-device result has L = max over shards (the reference's
    batch-level exit test is monotone), so shards are padded to the max L with their own further steps' logits being
    unavailable — callers that need exact single-device shapes in that mode run with max_length set (forced steps) or
    refine_iters >= 1, where L is always max_label_length + 1.  Here: L must match across ranks.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return logits
    logits = logits.contiguous()
    if uniform:
        out = torch.empty((world * logits.shape[0],) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
        dist.all_gather_into_tensor(out, logits, group=group)
        return out
    sizes = torch.tensor([logits.shape[0], logits.shape[1]], dtype=torch.int64, device=logits.device)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    bs = [int(s[0]) for s in all_sizes]
    if any(int(s[1]) != logits.shape[1] for s in all_sizes):
        raise RuntimeError('ranks produced different sequence lengths; use max_length=... or refine_iters >= 1 when sharding')
    bmax = max(bs)
    if logits.shape[0] < bmax:            # ragged last shards: pad to the largest shard, gather once, drop the padding
        pad = torch.zeros((bmax - logits.shape[0],) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
        logits = torch.cat([logits, pad], dim=0)
    out = torch.empty((world * bmax,) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    dist.all_gather_into_tensor(out, logits, group=group)
    if len(set(bs)) == 1:
        return out
    out = out.view(world, bmax, *logits.shape[1:])
    return torch.cat([out[r, :bs[r]] for r in range(world)], dim=0)


def data_parallel_forward(model, images: Tensor, max_length: Optional[int] = None, group=None) -> Tensor:
    """Every rank holds the same global batch `images` (or at least its own shard's rows); each computes its contiguous
    shard and all ranks return the full [N, L, C] logits."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(images.shape[0], world, rank)
    local = model(images[lo:hi], max_length)
    return all_gather_logits(local, group)


def average_gradients(flat: Tensor, group=None, bucket_elems: int = 6 * 1024 * 1024) -> Tensor:
    """Data-parallel gradient synchronisation of the training step (row N3): the mean over ranks of the flat gradient buffer,
    in place — what DDP's reducer leaves in `.grad` (the reference trains with Lightning's DDP strategy, train.py:88-96).

    The buffer already IS one contiguous bucket list (the library lays all 175 gradients out back to back), so there is no
    per-tensor flattening: it is all-reduced in slices of `bucket_elems` floats (24 MB by default — on an 8-GPU xGMI ring each
    link moves 2 * 7/8 of that per bucket, ~0.3 ms at 153 GB/s, and four buckets cover PARSeq-S), issued back to back on the
    collective stream so a later bucket's reduce-scatter overlaps the earlier one's all-gather."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    works = []
    for lo in range(0, flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[lo:lo + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    flat.mul_(1.0 / world)
    return flat
