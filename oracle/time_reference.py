#!/usr/bin/env python3
"""Times the REFERENCE's own `PARSeq.forward` (unmodified strhub model.py / modules.py from /root/reference, running on the timm
stand-in) beside the CPU oracle on the same inputs, weights, thread count and decode settings.   *** TEST INFRASTRUCTURE ***

Runs only in the build container (needs /root/reference).  Purpose: `bench.py`'s `cpu_baseline` times the oracle (kind "port")
because /root/reference does not exist on the GPU box; this script records, where both can run, that the port is neither
slower nor faster than the code it stands for (method of /root/reference/bench.py:43-49: repeated timed forwards).

Usage:  python oracle/time_reference.py [--ref /root/reference] [--batch 64] [--seconds 12] [--threads N]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import parseq_oracle as O  # noqa: E402
from oracle.make_golden import build_reference  # noqa: E402
from oracle.synth import CONFIGS, synth_images, synth_state_dict  # noqa: E402


def timed(fn, seconds):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 20:
            return n, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--seconds', type=float, default=12.0)
    ap.add_argument('--threads', type=int, default=min(os.cpu_count() or 1, 32))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    x = synth_images(args.batch, cfg, seed=1234)
    model, tok = build_reference(args.ref, cfg, sd)
    model.decode_ar, model.refine_iters = True, 1
    with torch.inference_mode():
        got_ref = model(tok, x, 25)
        got_orc = O.forward(sd, cfg, x, 25, decode_ar=True, refine_iters=1)
        n_r, dt_r = timed(lambda: model(tok, x, 25), args.seconds)
        n_o, dt_o = timed(lambda: O.forward(sd, cfg, x, 25, decode_ar=True, refine_iters=1), args.seconds)
    print(json.dumps({
        'workload': f'PARSeq-S fp32, batch {args.batch}, AR (26 steps forced) + 1 refinement, synthetic weights / crops, {args.threads} threads, '
                    f'host {os.cpu_count()} hardware threads',
        'reference_images_per_s': round(n_r * args.batch / dt_r, 2), 'reference_runs': n_r,
        'oracle_images_per_s': round(n_o * args.batch / dt_o, 2), 'oracle_runs': n_o,
        'max_abs_logit_difference': float((got_ref - got_orc).abs().max()),
        'note': 'reference = /root/reference strhub/models/parseq/model.py on oracle/timm_standin.py; oracle = oracle/parseq_oracle.py'}))


if __name__ == '__main__':
    main()
