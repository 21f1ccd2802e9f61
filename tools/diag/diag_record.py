"""Compare the training forward's record slot by slot: one launch vs per-operation launches (layout of lib_train.hip TrainEncoderLayout)."""
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from gpu_util import DEV, make_model
from parseq_amd import _native
from parseq_amd.train import _set_train_precision
from oracle.synth import CONFIGS, synth_images
cfg = CONFIGS['parseq']
m = make_model('parseq', 'bf16'); m.train_precision = 'bf16'
B = int(os.environ.get('DIAG_B', '8'))
images = synth_images(B, cfg, seed=23).to(DEV)
lib = _native.lib()
native = m.model._sync_native().model
_set_train_precision(m, native)
E, F, S, PK, depth = 384, 1536, 128, 96, 12
MS = B * S
r64 = lambda n: (n + 63) // 64 * 64
off = r64(MS * PK); layer0 = off
slots = {}
cur = 0
for name, n in [('x', MS * E), ('qkv', MS * 3 * E), ('ao', MS * E), ('x_mid', MS * E), ('hpre', MS * F), ('hact', MS * F), ('n1', MS * E), ('n2', MS * E)]:
    slots[name] = (cur, n); cur += r64(n)
stride = cur
x_last = layer0 + stride * depth
def fwd():
    nbytes = lib.parseq_train_encoder_workspace_bytes(native, B)
    ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=DEV)
    mem = torch.empty(B, S, E, dtype=torch.float32, device=DEV)
    _native.check(lib.parseq_train_encoder_forward(native, _native.ptr(images), B, _native.ptr(mem), _native.ptr(ws), nbytes, _native.stream_ptr(images)))
    torch.cuda.synchronize()
    return ws, mem
wa, ma = fwd()
os.environ['PARSEQ_TRAIN_ENC_PER_OP'] = '1'
wb, mb = fwd()
print('memory max abs diff', float((ma - mb).abs().max()), 'scale', float(mb.abs().max()))
def view(ws, l, name):
    o, n = slots[name]
    t = ws[layer0 + l * stride + o: layer0 + l * stride + o + n]
    if name in ('ao', 'hpre', 'hact', 'n1', 'n2'):
        t = t.view(torch.bfloat16)[:n].float()
    return t
for l in range(depth):
    row = []
    for name in slots:
        a, b = view(wa, l, name), view(wb, l, name)
        d = (a - b).abs()
        row.append(f'{name} {float(d.max()):.2e}/{float(b.abs().max()):.1e} (rms {float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()):.1e})')
    print(l, ' | '.join(row))
a, b = wa[x_last:x_last + MS * E], wb[x_last:x_last + MS * E]
print('x_last', float((a - b).abs().max()), float(b.abs().max()))
a, b = view(wa, 0, 'qkv').view(MS, 3 * E), view(wb, 0, 'qkv').view(MS, 3 * E)
bad = ((a - b).abs() > 2e-2).nonzero()
print('bad', bad.shape[0], 'of', a.numel())
rows, cols = bad[:, 0] % 128, bad[:, 1]
print('part (q/k/v) counts', [int(((cols // E) == u).sum()) for u in range(3)])
print('rows%128 unique', rows.unique().tolist()[:64])
print('cols unique', cols.unique().tolist()[:80])
print('cols%64 unique', (cols % 64).unique().tolist())
if bad.shape[0]:
    i = bad[0]; print('example', i.tolist(), float(a[i[0], i[1]]), float(b[i[0], i[1]]))
