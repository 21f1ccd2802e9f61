#!/usr/bin/env python3
"""CPU study: which CHEAPER operand formats than bf16 hi + lo on both sides (three matrix-core products) keep
PARSeq-S within the 1e-3 logit tolerance of the exact fp32 path?  No GPU: the oracle is run with roundings applied at named sites
(oracle/parseq_oracle.py `_r`), which is what each candidate's arithmetic amounts to:

  * "A hi+lo (fp16) x W fp16": two products per GEMM — equals exact activations times fp16-rounded weights   -> {'enc.w': 'fp16'}
  * "A fp16 x W hi+lo":        two products                                                                  -> {'enc.act': 'fp16'}
  * single fp16 product for the attention products only (q, k, v, p rounded to fp16)                         -> {'enc.qkv': 'fp16', 'enc.p': 'fp16'}
  * decoder K / V rows stored as fp16 / bf16 + 8-bit residual instead of f32 (the AR loop's HBM stream)       -> {'dec.kv': ...}

Reports, on N distinct seeded crops with the synthetic PARSeq-S weights: max |dlogit| of one NAR pass (no decision feedback) and of
the full AR(26) + 1 refinement forward, and the fraction of argmax-identical positions of the latter.
Usage: python tools/cheap_exact_study.py [--n 256] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import parseq_oracle as O            # noqa: E402
from oracle.synth import CONFIGS, synth_images, synth_state_dict      # noqa: E402

CANDIDATES = {
    'enc W fp16 (A hi+lo x W: 2 products)': {'enc.w': 'fp16'},
    'enc A fp16 (A x W hi+lo: 2 products)': {'enc.act': 'fp16', 'img': 'fp16'},
    'enc attention operands fp16 (1 product)': {'enc.qkv': 'fp16', 'enc.p': 'fp16'},
    'enc W fp16 + attention fp16': {'enc.w': 'fp16', 'enc.qkv': 'fp16', 'enc.p': 'fp16'},
    'enc fc1/fc2-only would be ~ 1/sqrt2 of "enc W fp16"': None,
    'enc everything fp16 (1 product)': {'enc.w': 'fp16', 'enc.act': 'fp16', 'img': 'fp16', 'enc.qkv': 'fp16', 'enc.p': 'fp16'},
    'dec K/V fp16': {'dec.kv': 'fp16'},
    'dec K/V bf16 + 8-bit residual': {'dec.kv': 'bf16+8'},
    'dec K/V bf16': {'dec.kv': 'bf16'},
    'dec K 24-bit, V bf16': {'dec.k': 'bf16+8', 'dec.v': 'bf16'},
    'dec K 24-bit, V fp16': {'dec.k': 'bf16+8', 'dec.v': 'fp16'},
    'dec K fp16, V 24-bit': {'dec.k': 'fp16', 'dec.v': 'bf16+8'},
    'dec W fp16 (2 products)': {'dec.w': 'fp16'},
    'enc W fp16 + attention fp16 + dec K/V fp16': {'enc.w': 'fp16', 'enc.qkv': 'fp16', 'enc.p': 'fp16', 'dec.kv': 'fp16'},
    'bf16 everywhere (the throughput mode)': 'bf16',
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--chunk', type=int, default=64)
    ap.add_argument('--json', default=None)
    ap.add_argument('--only', default=None, help='substring filter on candidate names')
    a = ap.parse_args()
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    images = synth_images(a.n, cfg, seed=1234)

    def run(rounding):
        nar, full = [], []
        with torch.inference_mode():
            for i in range(0, a.n, a.chunk):
                im = images[i:i + a.chunk]
                nar.append(O.forward(sd, cfg, im, None, decode_ar=False, refine_iters=0, rounding=rounding))
                full.append(O.forward(sd, cfg, im, 25, decode_ar=True, refine_iters=1, rounding=rounding))
        return torch.cat(nar), torch.cat(full)

    t0 = time.time()
    ref_nar, ref_full = run(None)
    print(f'# exact reference on {a.n} crops: {time.time() - t0:.0f} s', flush=True)
    out = {}
    print('| candidate | NAR max abs | AR+1 max abs | AR+1 argmax agree | rows with a differing string |')
    print('|---|---:|---:|---:|---:|')
    for name, rounding in CANDIDATES.items():
        if rounding is None or (a.only and a.only not in name):
            continue
        nar, full = run(rounding)
        e_nar = float((nar - ref_nar).abs().max())
        e_full = float((full - ref_full).abs().max())
        agree = float((full.argmax(-1) == ref_full.argmax(-1)).float().mean())
        rows = int((full.argmax(-1) != ref_full.argmax(-1)).any(-1).sum())
        out[name] = {'nar_max_abs': e_nar, 'ar1_max_abs': e_full, 'ar1_argmax_agree': agree, 'rows_differing': rows}
        print(f'| {name} | {e_nar:.2e} | {e_full:.2e} | {agree:.5f} | {rows} / {a.n} |', flush=True)
    if a.json:
        json.dump({'n': a.n, 'results': out}, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
