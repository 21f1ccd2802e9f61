#!/usr/bin/env python3
"""Sweep GEMM tile configurations of libparseq_hip on the encoder's shapes (run on the GPU box).
Usage: python tools/gemm_bench.py [--cfgs 0,1,2] [--iters 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parseq_amd import _native as nat  # noqa: E402

SHAPES = {  # name: (M, N, K, act)
    'fc1+gelu': (65536, 1536, 384, 1), 'fc1': (65536, 1536, 384, 0), 'fc2': (65536, 384, 1536, 0), 'qkv': (65536, 1152, 384, 0),
    'proj': (65536, 384, 384, 0), 'n1536k64': (65536, 1536, 64, 0), 'n384k64': (65536, 384, 64, 0), 'n1536k64g': (65536, 1536, 64, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfgs', default='0,100,3,103')
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--dtype', default='bf16')
    args = ap.parse_args()
    lib = nat.lib()
    tdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    code = nat.PARSEQ_BF16 if args.dtype == 'bf16' else nat.PARSEQ_F32
    for name, (M, N, K, act) in SHAPES.items():
        A = torch.randn(M, K, device='cuda').to(tdt)
        W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(tdt)
        bias = torch.randn(N, device='cuda')
        out = torch.empty(M, N, dtype=tdt if act else torch.float32, device='cuda')
        ref = None
        for cfg in [int(c) for c in args.cfgs.split(',')]:
            def run():
                nat.check(lib.parseq_op_linear_cfg(nat.ptr(A), nat.ptr(W), nat.ptr(bias), nat.ptr(out), code, act, M, N, K, cfg, nat.stream_ptr()))
            try:
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                print(f'{name:9s} cfg {cfg}: {e}')
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            chk = out.float().double().sum().item()
            ref = chk if ref is None else ref
            print(f'{name:9s} {M}x{N}x{K} cfg {cfg}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  checksum-delta {abs(chk - ref) / (abs(ref) + 1e-9):.1e}')


if __name__ == '__main__':
    main()
