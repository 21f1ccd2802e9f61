#!/usr/bin/env python3
"""Throughput of the training step (row N3, BASELINE.json configs[4]): PARSeq-S, 384 crops per GPU, K = 6 permutations,
forward + backward + gradient averaging across ranks + clip + AdamW, synthetic crops and labels resident on the device.

Not the headline metric (bench.py measures that); same protocol: W warm-up steps, K timed steps bracketed by a barrier and a
device synchronise, max over ranks, one JSON line from rank 0.  `--train-precision bf16` (default) rounds the operands of the
Linear products to bfloat16 (fp32 accumulate, fp32 master weights); `fp32` is the exact-product gate mode.

    python tools/train_bench.py [--batch 384] [--steps 3] [--warmup 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_bench.py --gpus 8
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=384, help='crops per GPU per step (configs/main.yaml:15)')
    ap.add_argument('--model', default='parseq')
    ap.add_argument('--train-precision', default='bf16', choices=['fp32', 'bf16'], help="GEMM operands of the step: exact fp32 products, or rounded to bf16 (fp32 accumulate / master weights)")
    args = ap.parse_args()
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    from parseq_amd import create_model
    from parseq_amd.train import TrainStep
    torch.manual_seed(0)
    system = create_model(args.model, precision='bf16').to(dev)
    system.train_precision = args.train_precision
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    ih, iw = system.hparams.img_size
    images = (torch.rand(B, 3, ih, iw, generator=g) * 2 - 1).to(dev)
    charset = system.hparams.charset_train
    lengths = torch.randint(1, 26, (B,), generator=g).tolist()
    lengths[0] = 25                                                  # the longest label sets the sequence length of the batch
    labels = [''.join(charset[int(i)] for i in torch.randint(0, len(charset), (n,), generator=g)) for n in lengths]
    step = TrainStep(system, total_steps=args.steps + args.warmup + 1, num_devices=world)

    def run(n):
        for _ in range(n):
            loss = step(images, labels)
        return loss

    run(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'training images/sec (32x128 crops) PARSeq-S, K=6 permutations, AdamW', 'value': round(world * B * args.steps / el, 1),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * el / args.steps, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.train_precision == 'fp32' else 'bf16', 'data': 'synthetic',
            'final_loss': round(float(loss), 4),
            'config': {'workload': f'{args.model} training step, batch={B}/GPU, labels of 1..25 characters (sequence length 26), 6 permutations, '
                                   f'dropout {system.hparams.dropout if system.training else 0} (decoder, 8 sites per pass), Linear products '
                                   f'{"exact fp32 on the f32 matrix cores" if args.train_precision == "fp32" else "with bf16 operands / fp32 accumulate / fp32 master weights"}, '
                                   f'attention / LayerNorm / loss / AdamW fp32 (BASELINE.json configs[4] trains bf16-mixed)',
                       'global_batch': world * B, 'parallelism': f'dp{world}' + (' + RCCL all-reduce of the flat gradient buffer' if world > 1 else '')}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
