"""Feasibility probe: one training forward + backward of 384 crops on one stream against two of 192 crops on two streams at once (no optimiser; the
two halves share the model's second stream, so their RESULTS are not to be trusted — this only asks whether the device has room to overlap them)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from gpu_util import DEV, make_model
from parseq_amd.train import loss_and_grads
from oracle.synth import CONFIGS, synth_images
from test_training import CHARSET_94
cfg = CONFIGS['parseq']
m = make_model('parseq', 'bf16'); m.train(); m.train_precision = 'bf16'
gen = torch.Generator().manual_seed(5)
B = 384
images = synth_images(B, cfg, seed=3).to(DEV)
lengths = torch.randint(1, 26, (B,), generator=gen).tolist(); lengths[0] = 25; lengths[B // 2] = 25
labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]
m.rng = np.random.default_rng(1)
perms = m.gen_tgt_perms(m.tokenizer.encode(labels, DEV))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def full():
    with torch.cuda.stream(s1):
        loss_and_grads(m, images, labels, perms)
def halves():
    with torch.cuda.stream(s1):
        loss_and_grads(m, images[:B // 2], labels[:B // 2], perms)
    with torch.cuda.stream(s2):
        loss_and_grads(m, images[B // 2:], labels[B // 2:], perms)
def halves_serial():
    with torch.cuda.stream(s1):
        loss_and_grads(m, images[:B // 2], labels[:B // 2], perms)
        loss_and_grads(m, images[B // 2:], labels[B // 2:], perms)
for name, fn in (('one batch of 384', full), ('two of 192 on two streams', halves), ('two of 192 one after the other', halves_serial), ('one batch of 384', full)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 6
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per 384 crops (forward + backward, no optimiser)')
