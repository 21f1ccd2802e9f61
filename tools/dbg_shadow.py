#!/usr/bin/env python3
"""Where do the bf16-shadow and the fp32-in-memory encoder forwards part?  (diagnostic; GPU)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from parseq_amd import _native, create_model
from parseq_amd.train import _set_train_precision

dev = torch.device('cuda', 0)
torch.manual_seed(0)
system = create_model('parseq', precision='bf16').to(dev)
system.train_precision = 'bf16'
model = system.model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
images = (torch.rand(B, 3, 32, 128, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
lib = _native.lib()
native = model._sync_native().model
_set_train_precision(system, native)
E, S = 384, 128
F = 4 * E
MS = B * S


def fwd():
    ws_bytes = lib.parseq_train_encoder_workspace_bytes(native, B)
    ws = torch.zeros(ws_bytes // 4, dtype=torch.float32, device=dev)
    memory = torch.empty(B, S, E, dtype=torch.float32, device=dev)
    _native.check(lib.parseq_train_encoder_forward(native, _native.ptr(images), B, _native.ptr(memory), _native.ptr(ws), ws_bytes, _native.stream_ptr(images)))
    torch.cuda.synchronize()
    return ws, memory


wa, ma = fwd()
os.environ['PARSEQ_TRAIN_NO_SHADOWS'] = '1'
wb, mb = fwd()
del os.environ['PARSEQ_TRAIN_NO_SHADOWS']
wc, mc = fwd()
print('memory: shadows vs fp32-in-memory max|d| =', float((ma - mb).abs().max()), ' shadows twice:', float((ma - mc).abs().max()))
layer0 = MS * 96
stride = MS * 16 * E
for i in range(12):
    base = layer0 + i * stride
    offs = {'x': (0, MS * E), 'qkv': (MS * E, MS * 3 * E), 'x_mid': (MS * 5 * E, MS * E), 'hpre': (MS * 6 * E, MS * F)}
    line = []
    for name, (o, n) in offs.items():
        a, b = wa[base + o: base + o + n], wb[base + o: base + o + n]
        line.append(f'{name} {float((a - b).abs().max()):.3e} ({int((a != b).sum())} of {n})')
    print(f'block {i}: ' + '  '.join(line))
    if i == 0:
        # the bf16 shadows against the fp32 copies rounded
        for name, o, n in (('n1', MS * (6 * E + 2 * F), MS * E), ('ao', MS * 4 * E, MS * E), ('hact', MS * (6 * E + F), MS * F), ('n2', MS * (7 * E + 2 * F), MS * E)):
            sh = wa[base + o: base + o + n // 2].view(torch.bfloat16)
            ref = wb[base + o: base + o + n].to(torch.bfloat16)
            print(f'   {name}: shadow vs rounded fp32: {int((sh != ref).sum())} of {n} differ, max {float((sh.float() - ref.float()).abs().max()):.3e}')
