#!/usr/bin/env python3
"""Times BUILDS of the product library against each other on the bf16x3 one-launch encoder alone (x3w::enc_blocks_x3w_kernel through the
parseq_op_enc_blocks_x3w hook): PARSeq-S shapes, batch 512, twelve blocks + tail, random weights; interleaved rounds, median and min per
build; the first build is the reference of a bit-for-bit comparison of the K | V rows (compiler-flag builds must be identical).

    python -m parseq_amd.build --variant <name> --units kern_enc_blocks_x3w <extra hipcc flags>      # parseq_amd/lib/libparseq_hip_<name>.so
    python tools/enc_x3w_build_ab.py product <name> [<name> ...]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parseq_amd import _native as nat   # noqa: E402

DEV = 'cuda'
E, F, depth, images = 384, 1536, 12, int(os.environ.get('X3_IMAGES', '512'))
M = images * 128
g = torch.Generator().manual_seed(0)
shapes = [(E,), (E,), (3 * E, E), (3 * E,), (E, E), (E,), (E,), (E,), (F, E), (F,), (E, F), (E,)]
tens, offs, total = [], [], 0
for _ in range(depth):
    for i, sh in enumerate(shapes):
        t = torch.randn(*sh, generator=g)
        t = t / sh[1] ** 0.5 if len(sh) == 2 else (1 + 0.1 * t if i in (0, 6) else 0.1 * t)
        tens.append(t); offs.append(total); total += (t.numel() + 31) // 32 * 32
for t in [1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g), torch.randn(2 * E, E, generator=g) / E ** 0.5, 0.1 * torch.randn(2 * E, generator=g)]:
    tens.append(t); offs.append(total); total += (t.numel() + 31) // 32 * 32
master = torch.zeros(total)
for t, o in zip(tens, offs):
    master[o:o + t.numel()] = t.reshape(-1)
md = master.to(DEV)
pack = torch.empty(total, dtype=torch.float32, device=DEV)
main = nat.lib()
nat.check(main.parseq_op_split_pack(nat.ptr(md), nat.ptr(pack), total, nat.stream_ptr()))
o32 = (C.c_uint32 * (12 * depth))(*offs[:12 * depth]); t32 = (C.c_uint32 * 4)(*offs[12 * depth:])
table = torch.empty(depth * 48, dtype=torch.uint8, device=DEV)
scratch = torch.empty(images * 393216 // 4, dtype=torch.float32, device=DEV)
x0 = torch.randn(M, E, generator=g).to(DEV)
kmem = torch.empty(images, 12, 128, 32, device=DEV); vmem = torch.empty_like(kmem)
SIG = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_void_p]
names = sys.argv[1:] or ['product']
fns = {}
for n in names:
    L = main if n == 'product' else C.CDLL(os.path.join(ROOT, 'parseq_amd', 'lib', f'libparseq_hip_{n}.so'))
    f = L.parseq_op_enc_blocks_x3w
    f.restype, f.argtypes = C.c_int, SIG
    fns[n] = f


def run(f):
    x = x0.clone()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = f(nat.ptr(x), nat.ptr(md), nat.ptr(pack), total, o32, depth, M, nat.ptr(table), nat.ptr(scratch), t32, nat.ptr(kmem), nat.ptr(vmem), nat.stream_ptr())
    b.record(); torch.cuda.synchronize()
    assert r == 0, r
    return a.elapsed_time(b)


times, same, ref = {n: [] for n in names}, {}, None
for rnd in range(int(os.environ.get('X3_ROUNDS', '9')) + 1):
    for n in names:
        t = run(fns[n])
        if rnd == 0:      # warm-up round: clocks, code; also the identity check
            cur = torch.stack([kmem, vmem]).clone()
            ref = cur if ref is None else ref
            same[n] = bool(torch.equal(cur, ref))
        else:
            times[n].append(t)
print(f'| build | median ms | min ms | K, V rows bit-identical to {names[0]} |\n|---|---:|---:|---|')
for n in names:
    t = sorted(times[n])
    print(f'| {n} | {t[len(t) // 2]:.3f} | {t[0]:.3f} | {same[n]} |')
