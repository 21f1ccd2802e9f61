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

# ---- round 6: products on the 2x-rate matrix formats ------------------------------------------------------------------
# MFMA cost in units of one bf16 16x16x32 issue per 16 x 16 x 32 of product: bf16x3 = 3.  v_mfma_i32_16x16x64_i8 and the MX-scaled fp8
# K = 128 forms run at twice the bf16 rate (MI355X_MICROARCH.md:389-393): an int8 / fp8 product costs 0.5.

def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def prod_bf16x3(x, w):
    """Calibration: what the shipped exact mode computes — hi.hi + hi.lo + lo.hi on bf16 pairs, fp32 accumulate (3 units)."""
    xh, wh = _bf16(x), _bf16(w)
    xl, wl = _bf16(x - xh), _bf16(w - wh)
    return xh @ wh.T + xh @ wl.T + xl @ wh.T


def _slices_i8(x, pow2):
    """Two int8 slices per element against a per-row scale: x ~ s (q1 + q2 / R), q1, q2 in [-127, 127] (pow2: s a power of two, q1 in
    [-64, 64], R = 256 — the integer combination is then a shift; otherwise s = rowmax / 127, R = 254)."""
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    if pow2:
        s = torch.exp2(torch.ceil(torch.log2(amax))) / 64.0
        ratio = 256.0
    else:
        s = amax / 127.0
        ratio = 254.0
    u = x / s
    q1 = torch.round(u)
    q2 = torch.clamp(torch.round((u - q1) * ratio), -127, 127)
    return q1, q2, s, ratio


def _prod_i8(x, w, pow2, four):
    a1, a2, sa, r = _slices_i8(x, pow2)
    w1, w2, sw, _ = _slices_i8(w, pow2)
    # the int32 accumulators hold exact integers (|sum| < 2^24 here, so the float32 matmuls below are exact too)
    hh = (a1 @ w1.T).double()
    cross = (a1 @ w2.T + a2 @ w1.T).double()
    acc = hh + cross / r
    if four:
        acc = acc + (a2 @ w2.T).double() / (r * r)
    return (acc * sa.double() * sw.double().T).float()


def _mx_fp8(x, block=32):
    """OCP MX e4m3: blocks of 32 along the contraction with a shared power-of-two scale (e8m0), elements e4m3."""
    sh = x.shape
    xb = x.reshape(*sh[:-1], sh[-1] // block, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)          # e4m3 max normal 448 = 1.75 x 2^8
    q = (xb / scale).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) * scale
    return q.reshape(sh)


def prod_bf16_fp8mx(x, w):
    """hi.hi in bf16 (1 unit) + the two correction terms hi.lo + lo.hi in MX-scaled fp8 (0.5 each): 2 units."""
    xh, wh = _bf16(x), _bf16(w)
    xl, wl = x - xh, w - wh
    return xh @ wh.T + _mx_fp8(xh) @ _mx_fp8(wl).T + _mx_fp8(xl) @ _mx_fp8(wh).T


def prod_bf16_fp8mx_lo16(x, w):
    """The same with the hi operands of the correction terms kept in bf16 (not expressible on one fp8 MFMA: the bound of the scheme)."""
    xh, wh = _bf16(x), _bf16(w)
    xl, wl = x - xh, w - wh
    return xh @ wh.T + xh @ _mx_fp8(wl).T + _mx_fp8(xl) @ wh.T


O.PRODUCT_EMULATIONS.update({
    'bf16x3': prod_bf16x3,
    'i8x3_pow2': lambda x, w: _prod_i8(x, w, True, False),
    'i8x3_full': lambda x, w: _prod_i8(x, w, False, False),
    'i8x4_full': lambda x, w: _prod_i8(x, w, False, True),
    'bf16+fp8mx': prod_bf16_fp8mx,
    'bf16+fp8mx(lo only)': prod_bf16_fp8mx_lo16,
})

CANDIDATES = {
    'r6 calibration: enc + dec Linear products as bf16x3 (3 units; the shipped mode)': {'enc.prod': 'bf16x3', 'dec.prod': 'bf16x3'},
    'r6 enc Linear: 2 int8 slices, power-of-two row scale, a1w1 + a1w2 + a2w1 (1.5 units)': {'enc.prod': 'i8x3_pow2'},
    'r6 enc Linear: 2 int8 slices, rowmax/127 scale, 3 products (1.5 units)': {'enc.prod': 'i8x3_full'},
    'r6 enc Linear: 2 int8 slices, rowmax/127 scale, 4 products (2 units)': {'enc.prod': 'i8x4_full'},
    'r6 enc + dec Linear: 2 int8 slices, rowmax/127 scale, 3 products': {'enc.prod': 'i8x3_full', 'dec.prod': 'i8x3_full'},
    'r6 enc Linear: bf16 hi.hi + MX-fp8 (hi.lo + lo.hi) (2 units)': {'enc.prod': 'bf16+fp8mx'},
    'r6 enc Linear: bf16 hi.hi + corrections with only the lo operand in MX-fp8 (bound of the scheme)': {'enc.prod': 'bf16+fp8mx(lo only)'},

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
