#!/usr/bin/env python3
"""The training step as 1 / 2 / 3 / 4 micro-batches that run at once (parseq_amd.train.TrainStep(micro_batches=n)): ms per step, batch 384."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from parseq_amd import create_model
from parseq_amd.train import TrainStep

dev = torch.device('cuda')
torch.manual_seed(0)
B = int(os.environ.get('BATCH', '384'))
g = torch.Generator().manual_seed(4321)
images = (torch.rand(B, 3, 32, 128, generator=g) * 2 - 1).to(dev)
out = {}
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 1, 2]:
    if B % n:
        continue
    system = create_model('parseq', precision='bf16').to(dev)
    system.train_precision = 'bf16'
    system.train()
    charset = system.hparams.charset_train
    gl = torch.Generator().manual_seed(7)
    lengths = torch.randint(1, 26, (B,), generator=gl).tolist(); lengths[0] = 25
    labels = [''.join(charset[int(i)] for i in torch.randint(0, len(charset), (k,), generator=gl)) for k in lengths]
    step = TrainStep(system, total_steps=40, micro_batches=n)
    for _ in range(3):
        step(images, labels)
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            loss = step(images, labels)
        torch.cuda.synchronize()
        runs.append(1e3 * (time.perf_counter() - t0) / 5)
    runs.sort()
    print(f'micro_batches={n}: {runs[1]:.2f} ms per step (min {runs[0]:.2f}, max {runs[2]:.2f}), {B / runs[1] * 1e3:.0f} images/s, loss {float(loss):.4f}', flush=True)
    out.setdefault(str(n), []).append(round(runs[1], 3))
    del step, system
print(json.dumps(out))
