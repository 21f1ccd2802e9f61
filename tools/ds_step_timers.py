#!/usr/bin/env python3
"""Where an AR step's two fused kernels spend their time (diagnostic).  Needs a library built with -DDS_TIMERS=1
(decoder_step.h):

    python -m parseq_amd.build --variant dst --units lib_decode -DDS_TIMERS=1
    PARSEQ_HIP_LIB=parseq_amd/lib/libparseq_hip_dst.so python tools/ds_step_timers.py [bf16x3|bf16]

Workgroup 0's wave 0 stamps s_memtime (100 MHz: 10 ns ticks) after each phase; the stamps come back in row 0's AR logits
(refine_iters = 0).  Printed: per phase, the median over steps 1..24 of the time spent IN the phase, in microseconds."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parseq_amd import create_model   # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
torch.manual_seed(0)
model = create_model('parseq', decode_ar=True, refine_iters=0, precision=prec).eval().to('cuda')
x = (torch.rand(512, 3, 32, 128) * 2 - 1).to('cuda')
if prec == 'bf16':
    x = x.bfloat16()
with torch.inference_mode():
    for _ in range(3):
        logits = model(x, 25)
torch.cuda.synchronize()
row = logits[0].float().cpu()             # [26][C]: position p holds the stamps of the mid kernel that finished step p (and of the mlp kernel of step p)
MID = ['partial sums in LDS', 'decoder.norm', 'head + logits', 'pick', 'table self-attention', 'out_proj + residual', 'norm1', 'q-projection (to the end)']
MLP = ['ca / t staged', 'cross out_proj + residual', 'norm2', 'linear1 + GELU', 'linear2 partial (to the end)']
TICK_US = 0.01
steps = range(1, 25)


def phases(cols, names):
    st = torch.stack([row[p, cols[0]:cols[0] + len(names)] for p in steps])     # cumulative ticks
    d = torch.cat([st[:, :1], st[:, 1:] - st[:, :-1]], dim=1) * TICK_US
    med = d.median(dim=0).values
    for n, v in zip(names, med):
        print(f'  {n:34s} {float(v):6.2f} us')
    print(f'  {"(in-kernel total)":34s} {float(st[:, -1].median()) * TICK_US:6.2f} us')


print(f'precision {prec}, batch 512, workgroup 0; medians over AR steps 1..24')
print('dec_step_mid_kernel:')
phases((0,), MID)
print('dec_step_mlp_kernel:')
phases((16,), MLP)
