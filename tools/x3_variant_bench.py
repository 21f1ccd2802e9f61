#!/usr/bin/env python3
"""Times builds of the bf16x3 one-launch encoder (parseq_amd/lib/x3v/*.so, tools/x3_variants.sh) against each other on one GPU:
PARSeq-S shapes, batch 512, 12 blocks + tail, random weights; interleaved rounds, median and min per build; the first build
is the reference for a max-abs difference of the K rows (ablation builds are wrong by construction: the column says by how much)."""
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parseq_amd import _native as nat   # noqa: E402

lib = nat.lib()
DEV = 'cuda'
E, F, depth, images = 384, 1536, int(os.environ.get('X3_DEPTH', '12')), int(os.environ.get('X3_IMAGES', '512'))
M = images * 128
g = torch.Generator().manual_seed(0)
shapes = [(E,), (E,), (3 * E, E), (3 * E,), (E, E), (E,), (E,), (E,), (F, E), (F,), (E, F), (E,)]
tens, offs, total = [], [], 0
for l in range(depth):
    for i, sh in enumerate(shapes):
        t = torch.randn(*sh, generator=g)
        if len(sh) == 2:
            t = t / sh[1] ** 0.5
        elif i in (0, 6):
            t = 1 + 0.1 * t
        else:
            t = 0.1 * t
        tens.append(t); offs.append(total); total += (t.numel() + 31) // 32 * 32
tail = [1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g), torch.randn(2 * E, E, generator=g) / E ** 0.5, 0.1 * torch.randn(2 * E, generator=g)]
for t in tail:
    tens.append(t); offs.append(total); total += (t.numel() + 31) // 32 * 32
master = torch.zeros(total)
for t, o in zip(tens, offs):
    master[o:o + t.numel()] = t.reshape(-1)
md = master.to(DEV)
pack = torch.empty(total, dtype=torch.float32, device=DEV)
nat.check(lib.parseq_op_split_pack(nat.ptr(md), nat.ptr(pack), total, nat.stream_ptr()))
o32 = (C.c_uint32 * (12 * depth))(*offs[:12 * depth]); t32 = (C.c_uint32 * 4)(*offs[12 * depth:])
table = torch.empty(depth * 48, dtype=torch.uint8, device=DEV)
scratch = torch.empty(images * 393216 // 4, dtype=torch.float32, device=DEV)
x0 = torch.randn(M, E, generator=g).to(DEV)
kmem = torch.empty(images, 12, 128, 32, device=DEV); vmem = torch.empty_like(kmem)
TIMERS = '--timers' in sys.argv       # the named build was compiled with -DX3_TIMERS=1: print its per-phase s_memtime ticks (encoder_blocks_x3.h X3_TIMERS)
if TIMERS:
    sys.argv.remove('--timers')
names = [a for i, a in enumerate(sys.argv[1:]) if a != '--sustained' and a != '--timers8' and (i == 0 or sys.argv[i] != '--sustained')] or sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(ROOT, 'parseq_amd/lib/x3v/*.so')))
libs = {}


class _Product:
    """`prod_w8` / `prod_w4`: the product library's own kernels through its test hooks (parseq_op_enc_blocks_x3w / _x3: the same argument list as x3_variant_run)."""

    def __init__(self, fn):
        self.x3_variant_run = fn


for n in names:
    if n in ('prod_w8', 'prod_w4'):
        libs[n] = _Product(lib.parseq_op_enc_blocks_x3w if n == 'prod_w8' else lib.parseq_op_enc_blocks_x3)
        continue
    L = C.CDLL(os.path.join(ROOT, 'parseq_amd/lib/x3v', n + '.so'))
    L.x3_variant_run.restype = C.c_int
    L.x3_variant_run.argtypes = [C.c_void_p] * 3 + [C.c_longlong, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)] + [C.c_void_p] * 3
    libs[n] = L


def run(L):
    x = x0.clone()
    r = L.x3_variant_run(nat.ptr(x), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32, nat.ptr(kmem), nat.ptr(vmem), nat.stream_ptr())
    assert r == 0, r


if '--sustained' in sys.argv:      # N launches of the first build back to back (events between them, one synchronise at the end), then again after an idle
    import time
    k = sys.argv.index('--sustained'); N = int(sys.argv[k + 1]); names = [n for n in names if n not in ('--sustained', sys.argv[k + 1])]
    L = libs[names[0]] if names and names[0] in libs else list(libs.values())[0]

    def burst(n):
        x = x0.clone(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            assert L.x3_variant_run(nat.ptr(x), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32, nat.ptr(kmem), nat.ptr(vmem), nat.stream_ptr()) == 0
            ev[i + 1].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    burst(3)
    time.sleep(3.0)
    t = burst(N)
    print('after 3 s idle, launches back to back, ms:', ' '.join(f'{i}:{t[i]:.2f}' for i in sorted(set([0, 1, 2, 3, 5, 10, 20, 40, 80, N - 1])) if i < N))
    print(f'first 3 mean {sum(t[:3]) / 3:.3f}   last 10 mean {sum(t[-10:]) / 10:.3f}   min {min(t):.3f}  max {max(t):.3f}')
    time.sleep(3.0)
    t2 = burst(5)
    print('after another 3 s idle:', ' '.join(f'{v:.2f}' for v in t2))
    sys.exit(0)
times = {n: [] for n in names}
ref = None
diffs = {}
for rnd in range(int(os.environ.get('X3_ROUNDS', '5'))):
    for n in names:
        run(libs[n]); torch.cuda.synchronize()
        if rnd == 0:
            if ref is None:
                ref = torch.stack([kmem, vmem]).clone()
            diffs[n] = float((torch.stack([kmem, vmem]) - ref).abs().max())
        x = x0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L = libs[n]
        r = L.x3_variant_run(nat.ptr(x), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32, nat.ptr(kmem), nat.ptr(vmem), nat.stream_ptr())
        b.record(); torch.cuda.synchronize()
        times[n].append(a.elapsed_time(b))
print(f'| build | median ms | min ms | max|dK,dV| vs {names[0]} |\n|---|---:|---:|---:|')
for n in names:
    t = sorted(times[n])
    print(f'| {n} | {t[len(t) // 2]:.3f} | {t[0]:.3f} | {diffs[n]:.3e} |')

if '--timers8' in sys.argv[1:] or os.environ.get('X3W_TIMERS'):
    SL8 = ['parameters + LayerNorm', 'park / unpark / O reload', 'head loop: wait + barrier', 'head loop: MFMA pairs', 'head loop: epilogues + soft-max section', 'proj: wait + barrier', 'proj: MFMAs',
           'MLP: wait + barrier', 'MLP: GELU in front of a group (half B)', 'MLP: fc1 MFMAs', 'MLP: fc2 MFMAs', 'MLP: GELU behind a group (half A)', 'tail + rest']
    n = names[-1]
    x = x0.clone()
    r = libs[n].x3_variant_run(nat.ptr(x), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32, nat.ptr(kmem), nat.ptr(vmem), nat.stream_ptr())
    torch.cuda.synchronize()
    t = x.view(torch.int64).view(M, E // 2)[torch.arange(images * 8, device=DEV) * 16][:, :13].double().view(images, 8, 13)      # one row of 13 counters per wave
    for half, sel in (('half A (waves 0-3)', slice(0, 4)), ('half B (waves 4-7)', slice(4, 8))):
        th = t[:, sel].reshape(-1, 13); tot = th.sum(1)
        print(f'\n| phase of `enc_blocks_x3w_kernel` (build {n}, {half}, mean over {th.shape[0]} waves) | s_memtime ticks per wave | share |\n|---|---:|---:|')
        for i, name in enumerate(SL8):
            print(f'| {name} | {th[:, i].mean():,.0f} | {100 * th[:, i].sum() / tot.sum():.1f} % |')
        print(f'| total | {tot.mean():,.0f} | (min {tot.min():,.0f}, max {tot.max():,.0f}) |')

if TIMERS:
    SLOTS = ['parameters + LayerNorm 1', 'park', 'head loop: q / k / v pairs', 'head loop: q / k / v epilogues', 'head loop: S, soft-max, P V, O stores', 'unpark + O reload', 'proj',
             'parameters + LayerNorm 2', 'MLP: fc1 pairs', 'MLP: GELU blocks', 'MLP: fc2 pairs', 'tail (final LayerNorm + K | V)', 'biases / rest']
    n = names[-1]
    x = x0.clone()
    r = libs[n].x3_variant_run(nat.ptr(x), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32, nat.ptr(kmem), nat.ptr(vmem), nat.stream_ptr())
    torch.cuda.synchronize()
    t = x.view(torch.int64).view(M, E // 2)[torch.arange(images * 4, device=DEV) * 32][:, :13].double()      # one row of 13 counters per wave
    tot = t.sum(1)
    print(f'\n| phase of `enc_blocks_x3_kernel` (build {n}, mean over {t.shape[0]} waves) | s_memtime ticks per wave | share |\n|---|---:|---:|')
    for i, name in enumerate(SLOTS):
        print(f'| {name} | {t[:, i].mean():,.0f} | {100 * t[:, i].sum() / tot.sum():.1f} % |')
    print(f'| total | {tot.mean():,.0f} | (min {tot.min():,.0f}, max {tot.max():,.0f}) |')
