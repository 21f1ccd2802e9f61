#!/usr/bin/env python3
"""Ablations of the LN-panel GEMM kernel (fc1 shape 65536 x 1536 x 384 and qkv-width 1152) — run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parseq_amd import _native as nat  # noqa: E402

NAMES = {0: 'full', 1: 'no global stores', 2: 'no LayerNorm prologue', 3: 'no MFMA / LDS reads', 4: 'no vmcnt waits (garbage)'}


def main():
    lib = nat.lib()
    M, E = 65536, 384
    x = torch.randn(M, E, device='cuda')
    gamma, beta = torch.rand(E, device='cuda') + 0.5, torch.randn(E, device='cuda') * 0.1
    for N in (1536, 1152):
        W = (torch.randn(N, E, device='cuda') / E ** 0.5).bfloat16()
        bias = torch.randn(N, device='cuda') * 0.1
        out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
        for v in range(5):
            def run():
                nat.check(lib.parseq_op_ln_linear_gelu(nat.ptr(x), nat.ptr(gamma), nat.ptr(beta), nat.ptr(W), nat.ptr(bias), nat.ptr(out), M, N, v, nat.stream_ptr()))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            print(f'N={N} variant {v} ({NAMES[v]:26s}): {us:8.1f} us  {2.0 * M * N * E / us / 1e6:7.1f} TFLOP/s-equivalent')
        if N == 1536:
            ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x[:256], (E,), gamma, beta, 1e-6).bfloat16().float() @ W.float().T + bias)
            nat.check(lib.parseq_op_ln_linear_gelu(nat.ptr(x), nat.ptr(gamma), nat.ptr(beta), nat.ptr(W), nat.ptr(bias), nat.ptr(out), M, N, 0, nat.stream_ptr()))
            torch.cuda.synchronize()
            print('max |err| vs torch on 256 rows:', (out[:256].float() - ref).abs().max().item())


def mlp():
    lib = nat.lib()
    M, E, F = 65536, 384, 1536
    names = {0: 'full', 1: 'no weight stream in loop', 2: 'no GELU', 3: 'no LDS reads / MFMA', 4: 'no barriers / waits', 5: 'no LayerNorm prologue'}
    x = torch.randn(M, E, device='cuda')
    gamma, beta = torch.rand(E, device='cuda') + 0.5, torch.randn(E, device='cuda') * 0.1
    W1 = (torch.randn(F, E, device='cuda') / E ** 0.5).bfloat16(); W2 = (torch.randn(E, F, device='cuda') / F ** 0.5).bfloat16()
    b1, b2 = torch.randn(F, device='cuda') * 0.1, torch.randn(E, device='cuda') * 0.1
    for v in range(6):
        def run():
            nat.check(lib.parseq_op_mlp_variant(nat.ptr(x), nat.ptr(gamma), nat.ptr(beta), nat.ptr(W1), nat.ptr(b1), nat.ptr(W2), nat.ptr(b2), M, v, nat.stream_ptr()))
        x.normal_()
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.normal_()
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 10
        print(f'fused MLP variant {v} ({names[v]:26s}): {us:8.1f} us  {4.0 * M * E * F / us / 1e6:7.1f} TFLOP/s-equivalent')


def mlp_stamps():
    lib = nat.lib()
    M, E, F = 65536, 384, 1536
    xbuf = torch.randn(M * E + 1024, device='cuda')          # 4 KB of stamp space behind the matrix
    gamma, beta = torch.rand(E, device='cuda') + 0.5, torch.randn(E, device='cuda') * 0.1
    W1 = (torch.randn(F, E, device='cuda') / E ** 0.5).bfloat16(); W2 = (torch.randn(E, F, device='cuda') / F ** 0.5).bfloat16()
    b1, b2 = torch.randn(F, device='cuda') * 0.1, torch.randn(E, device='cuda') * 0.1
    for _ in range(3):
        nat.check(lib.parseq_op_mlp_variant(nat.ptr(xbuf), nat.ptr(gamma), nat.ptr(beta), nat.ptr(W1), nat.ptr(b1), nat.ptr(W2), nat.ptr(b2), M, 6, nat.stream_ptr()))
    torch.cuda.synchronize()
    st = xbuf[M * E:].view(torch.int64).cpu().view(-1)[:8 * 64].view(2, 4, 64)
    if 'fine' in sys.argv:      # library built with -DMLP_FINE_STAMPS: a stamp at every stage start of chunks 0 and 1
        names = ['start', 'LN done'] + [f'c0t{t}' for t in range(6)] + ['chunk1'] + [f'c1t{t}' for t in range(6)] + ['chunk2', 'last chunk', 'loop done', 'epilogue done']
        for blk in range(2):
            v = st[blk, 0, :len(names)].tolist()
            print(f'block {"0" if blk == 0 else "300"} wave 0: ' + ', '.join(f'{n} {(b - v[0]) / 2400.0:.2f}' for n, b in zip(names, v)))
        return
    names = ['start', 'LN done', 'chunk1', 'chunk2', 'last chunk', 'loop done', 'epilogue done']
    for blk in range(2):
        for w in range(4):
            v = st[blk, w, :7].tolist()
            print(f'block {"0" if blk == 0 else "300"} wave {w}: ' + ', '.join(f'{n} {(b - v[0]) / 2400.0:.1f}us' for n, b in zip(names, v)) + '   (s_memtime ticks / 2.4 GHz)')


def mlp_ab(variants=(0, 10), rounds=6, reps=10):
    """Interleaved A/B of fused-MLP variants in one process (cdna_hip_programming.md rule 24): median and min per variant."""
    lib = nat.lib()
    M, E, F = 65536, 384, 1536
    x = torch.randn(M, E, device='cuda')
    gamma, beta = torch.rand(E, device='cuda') + 0.5, torch.randn(E, device='cuda') * 0.1
    W1 = (torch.randn(F, E, device='cuda') / E ** 0.5).bfloat16(); W2 = (torch.randn(E, F, device='cuda') / F ** 0.5).bfloat16()
    b1, b2 = torch.randn(F, device='cuda') * 0.1, torch.randn(E, device='cuda') * 0.1
    times = {v: [] for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            x.normal_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                nat.check(lib.parseq_op_mlp_variant(nat.ptr(x), nat.ptr(gamma), nat.ptr(beta), nat.ptr(W1), nat.ptr(b1), nat.ptr(W2), nat.ptr(b2), M, v, nat.stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
            if r:                                             # round 0 is warm-up
                times[v].append(1e3 * e0.elapsed_time(e1) / reps)
    for v in variants:
        t = sorted(times[v])
        print(f'fused MLP variant {v:2d}: median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us  max {t[-1]:7.1f} us   '
              f'{4.0 * M * E * F / t[len(t) // 2] / 1e6:7.1f} TFLOP/s at the median')


def attn_fused():
    """Fused attention-branch kernel at batch 512 (M = 65536): time per launch and the phase stamps of two workgroups."""
    lib = nat.lib()
    M, E = 65536, 384
    xbuf = torch.randn(M * E + 1024, device='cuda')
    gamma, beta = torch.rand(E, device='cuda') + 0.5, torch.randn(E, device='cuda') * 0.1
    Wqkv = (torch.randn(3 * E, E, device='cuda') / E ** 0.5).bfloat16(); Wproj = (torch.randn(E, E, device='cuda') / E ** 0.5).bfloat16()
    bqkv, bproj = torch.randn(3 * E, device='cuda') * 0.1, torch.randn(E, device='cuda') * 0.1
    def run(v):
        nat.check(lib.parseq_op_attn_fused(nat.ptr(xbuf), nat.ptr(gamma), nat.ptr(beta), nat.ptr(Wqkv), nat.ptr(bqkv), nat.ptr(Wproj), nat.ptr(bproj), M, v, nat.stream_ptr()))
    times = []
    for r in range(6):
        xbuf.normal_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(0)
        e1.record()
        torch.cuda.synchronize()
        if r:
            times.append(1e3 * e0.elapsed_time(e1) / 10)
    t = sorted(times)
    fl = 2.0 * M * 4 * E * E + 4.0 * (M // 128) * 6 * 128 * 128 * 64
    print(f'fused attention branch: median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us  max {t[-1]:7.1f} us   {fl / t[len(t) // 2] / 1e6:7.1f} TFLOP/s at the median')
    xbuf.normal_()
    for _ in range(3):
        run(6)
    torch.cuda.synchronize()
    st = xbuf[M * E:].view(torch.int64).cpu().view(-1)[:8 * 64].view(2, 4, 64)
    names = ['start', 'LN done', 'head 1', 'last head', 'loop done', 'epilogue done']
    for blk in range(2):
        for w in range(4):
            v = st[blk, w, :len(names)].tolist()
            print(f'block {"0" if blk == 0 else "300"} wave {w}: ' + ', '.join(f'{n} {(b - v[0]) / 2400.0:.1f}us' for n, b in zip(names, v)) + '   (s_memtime ticks / 2.4 GHz)')


if __name__ == '__main__':
    if 'attn' in sys.argv:
        attn_fused()
    elif 'ab' in sys.argv:
        mlp_ab()
    elif 'stamps' in sys.argv:
        mlp_stamps()
    elif 'mlp' in sys.argv:
        mlp()
    else:
        main()
