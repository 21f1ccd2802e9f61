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


def all_gather_logits(logits: Tensor, group=None, uniform: bool = False, force: bool = False) -> Tensor:
    """[b_local, L, C] on every rank -> [sum b_local, L, C] on every rank (rank order).

    uniform=True: the caller guarantees identical shapes on every rank (fixed per-GPU batch, forced steps or
    refine_iters >= 1) — ONE all-gather and no host synchronisation, which is what a throughput loop wants.  Otherwise
    the shapes are exchanged first (a second small collective and a device->host read).

    Shards may differ in batch size (ragged last shard).  L must match across ranks here; the one mode in which shards can
    disagree on L (AR, refine_iters == 0, max_length=None: each shard may reach "every row holds an EOS" earlier than the
    whole batch would) is resolved BEFORE the gather by `data_parallel_forward`, which agrees on the single-device length
    with a 1-int all-reduce(max).
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not force:          # force: run the collective anyway (bench.py --force-dist, a one-rank self-test)
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
        raise RuntimeError('ranks produced different sequence lengths: shard with data_parallel_forward (which agrees on the '
                           'early-exit length first), or use max_length=... / refine_iters >= 1')
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
    with_len = getattr(model, 'forward_with_length', None)
    if with_len is None or world == 1:
        return all_gather_logits(model(images[lo:hi], max_length), group)
    # SURVEY.md section 8(e) option (ii).  The reference stops the AR loop at the first step after which EVERY row of the batch
    # holds an EOS (model.py:144-145) and returns that many positions.  A shard reaches the condition no later than the whole
    # batch does and the condition is monotone in the step, so the single-device length is the maximum of the shards' lengths:
    # every shard computes all positions anyway (no per-step host sync on the device), reports its own length, one 1-int
    # all-reduce(max) agrees on L, and every shard contributes logits[:, :L] — shape- and value-identical to a single-device run.
    full, length = with_len(images[lo:hi], max_length)
    agreed = torch.tensor([length], dtype=torch.int32, device=full.device)
    dist.all_reduce(agreed, op=dist.ReduceOp.MAX, group=group)
    return all_gather_logits(full[:, :int(agreed.item())], group)


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


def average_gradient_segments(flat: Tensor, segments, group=None, wait=None, comm_stream=None, force: bool = False) -> Tensor:
    """`average_gradients` overlapped with the backward that is still producing `flat` (DDP's bucketed reducer, reference
    train.py:65-71): `segments` = [(begin, end, event)] in the order the ranges of the flat gradient buffer become final
    (parseq_train_grad_segment).  For each one, `wait(comm_stream, event)` makes the side stream wait for the segment's event
    (parseq_stream_wait_event) and the segment's all-reduce is enqueued with that stream current — the collective's own stream
    starts behind it, i.e. as soon as the bucket is final, while the main stream is still running the earlier blocks' backward.
    The caller's current stream then waits for every collective and scales by 1 / world.  With `wait` / `comm_stream` None (CPU
    tensors under gloo in the tests) the segments are reduced in order on the spot.  `force`: run the collectives in a one-rank group too."""
    import contextlib

    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return flat
    segments = list(segments)
    if not segments:
        raise ValueError('no gradient segments')
    if comm_stream is not None and (wait is None or any(ev is None for _, _, ev in segments)):
        # without the event dependency the side stream would start reducing a range the backward on the main stream is still writing
        raise ValueError('comm_stream needs `wait` and an event for every segment')
    covered = sorted((b, e) for b, e, _ in segments)
    if covered[0][0] != 0 or covered[-1][1] != flat.numel() or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
        raise ValueError('gradient segments do not tile the flat buffer')
    works = []
    for begin, end, event in segments:
        ctx = contextlib.nullcontext()
        if comm_stream is not None:
            wait(comm_stream, event)
            ctx = torch.cuda.stream(comm_stream)
        with ctx:
            works.append(dist.all_reduce(flat[begin:end], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    if world > 1:
        flat.mul_(1.0 / world)
    return flat
