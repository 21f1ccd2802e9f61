#!/usr/bin/env python3
"""Where a training step's time goes, phase by phase (row N3): HIP events between the three library calls of a step — encoder forward,
decoder forward + backward (all K permutation passes), encoder backward — and the clip + AdamW tail, plus the wall time of the whole
step with nothing waiting in between.  Batch 384, PARSeq-S, bf16-operand mode, K = 6, dropout 0.1.

    python tools/diag/train_phases.py [--batch 384] [--steps 5]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=384)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    from parseq_amd import _native, create_model
    from parseq_amd import train as T
    dev = torch.device('cuda')
    torch.manual_seed(0)
    system = create_model('parseq', precision='bf16').to(dev)
    system.train_precision = 'bf16'
    system.train()
    g = torch.Generator().manual_seed(4321)
    B = a.batch
    images = (torch.rand(B, 3, 32, 128, generator=g) * 2 - 1).to(dev)
    charset = system.hparams.charset_train
    lengths = torch.randint(1, 26, (B,), generator=g).tolist()
    lengths[0] = 25
    labels = [''.join(charset[int(i)] for i in torch.randint(0, len(charset), (n,), generator=g)) for n in lengths]
    step = T.TrainStep(system, total_steps=a.steps * 3 + 8)
    for _ in range(2):
        step(images, labels)
    torch.cuda.synchronize()
    # whole steps, nothing in between
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(images, labels)
    torch.cuda.synchronize()
    whole = 1e3 * (time.perf_counter() - t0) / a.steps

    # the same step with events between the library calls (wraps the three entry points the step makes)
    lib = _native.lib()
    marks, hmarks = [], []
    names = ['parseq_train_encoder_forward', 'parseq_train_decoder', 'parseq_train_encoder_backward', 'parseq_grad_norm', 'parseq_adamw_step']
    originals = {n: getattr(lib, n) for n in names}

    class Wrapped:
        def __init__(self, fn, name):
            self.fn, self.name = fn, name

        def __call__(self, *args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0 = time.perf_counter()
            e0.record()
            rc = self.fn(*args)
            e1.record()
            marks.append((self.name, e0, e1))
            hmarks.append((self.name, h0, time.perf_counter()))
            return rc
    for n in names:
        setattr(lib, n, Wrapped(originals[n], n))
    try:
        host, host_tl = [], []
        for _ in range(a.steps):
            torch.cuda.synchronize()          # the device idle at the start of the step: host timestamps of the calls are then the host's own pace
            del hmarks[:]
            h0 = time.perf_counter()
            step(images, labels)
            host.append(1e3 * (time.perf_counter() - h0))
            host_tl.append([(n, round(1e3 * (b - h0), 3), round(1e3 * (e - h0), 3)) for n, b, e in hmarks])
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(lib, n, originals[n])
    phases = {n: 0.0 for n in names}
    gaps = 0.0
    for i, (n, e0, e1) in enumerate(marks):
        phases[n] += e0.elapsed_time(e1) / a.steps
        if i + 1 < len(marks):
            gaps += max(0.0, e1.elapsed_time(marks[i + 1][1])) / a.steps
    out = {'batch': B, 'steps': a.steps, 'ms_per_step_whole': round(whole, 3), 'host_enqueue_ms_per_step': round(sum(host) / len(host), 3),
           'phases_ms': {k: round(v, 3) for k, v in phases.items()}, 'between_calls_ms': round(gaps, 3),
           'sum_ms': round(sum(phases.values()) + gaps, 3),
           'host_timeline_last_step_ms (call, entered, returned; device idle at t = 0)': host_tl[-1]}
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
