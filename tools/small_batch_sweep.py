#!/usr/bin/env python3
"""Where does the one-launch encoder (one workgroup per image) stop paying?  Latency of one forward at small batches, the one-launch
encoder against the per-operation kernels (PARSEQ_NO_FUSED_X3 / PARSEQ_NO_FUSED_BLOCKS, read when a plan is created), one call at a
time — the call pattern of read.py and of the reference's bench.py.

    python tools/small_batch_sweep.py [--precision bf16x3] [--batches 1 2 4 8 16 32 64 128 256] [--json out.json]

Prints one row per batch: ms per forward (median of `--repeats` timed regions of `--steps` calls) for AR (26 steps) + 1 refinement and
for NAR + 3 refinements (the reference's published operating point), both encoder forms, and the max |difference| of their logits.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCH = {'bf16x3': 'PARSEQ_NO_FUSED_X3', 'bf16': 'PARSEQ_NO_FUSED_BLOCKS'}


class Form:
    """A model whose plans are all created under `env` (the library reads its switches when a plan is created, and a plan is re-created
    whenever the batch outgrows it: the environment is applied around EVERY call)."""

    def __init__(self, name, precision, sd, env):
        from parseq_amd import create_model
        self.env = env
        m = create_model(name, precision=precision)
        m.model.load_state_dict(sd)
        self.m = m.eval().to('cuda')

    def __enter__(self):
        self.keep = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        return self.m

    def __exit__(self, *a):
        for k, v in self.keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def time_calls(m, x, max_length, steps, repeats):
    runs = []
    with torch.inference_mode():
        for _ in range(3):
            out = m(x, max_length)
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = m(x, max_length)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / steps)
    return 1e3 * statistics.median(runs), out.float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='parseq')
    ap.add_argument('--precision', default='bf16x3', choices=list(SWITCH))
    ap.add_argument('--batches', type=int, nargs='+', default=[1, 2, 4, 8, 16, 32, 64, 128, 256])
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=5)
    ap.add_argument('--env', nargs='*', default=[], help='extra KEY=VALUE for BOTH forms (e.g. a threshold override)')
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    from parseq_amd import create_model
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in create_model(args.model).model.state_dict().items()}
    extra = dict(kv.split('=', 1) for kv in args.env)
    # the small-batch route itself is pinned off (0) / forced on by its own switch, so that the two forms stay what they say
    forms = {'one_launch': Form(args.model, args.precision, sd, {**extra, 'PARSEQ_SMALL_BATCH': '0'}),
             'per_op': Form(args.model, args.precision, sd, {**extra, SWITCH[args.precision]: '1'})}
    rows = []
    g = torch.Generator().manual_seed(7)
    for B in args.batches:
        x = (torch.rand(B, 3, *forms['one_launch'].m.hparams.img_size, generator=g) * 2 - 1).to('cuda')
        x = x.bfloat16() if args.precision == 'bf16' else x
        row = {'batch': B}
        for mode, ar, ri, ml in (('ar1', True, 1, 25), ('nar3', False, 3, None)):
            outs = {}
            for form, f in forms.items():
                with f as m:
                    m.model.decode_ar, m.model.refine_iters = ar, ri
                    ms, outs[form] = time_calls(m, x, ml, args.steps, args.repeats)
                row[f'{mode}_{form}_ms'] = round(ms, 4)
            row[f'{mode}_max_abs_between_forms'] = float((outs['one_launch'] - outs['per_op']).abs().max())
            row[f'{mode}_argmax_equal'] = bool(torch.equal(outs['one_launch'].argmax(-1), outs['per_op'].argmax(-1)))
        rows.append(row)
        print(json.dumps(row), flush=True)
    print('| batch | AR+1 one-launch ms | AR+1 per-op ms | NAR+3 one-launch ms | NAR+3 per-op ms | img/s best AR+1 |')
    print('|---:|---:|---:|---:|---:|---:|')
    for r in rows:
        best = min(r['ar1_one_launch_ms'], r['ar1_per_op_ms'])
        print(f"| {r['batch']} | {r['ar1_one_launch_ms']:.3f} | {r['ar1_per_op_ms']:.3f} | {r['nar3_one_launch_ms']:.3f} | {r['nar3_per_op_ms']:.3f} | {1e3 * r['batch'] / best:.0f} |")
    if args.json:
        json.dump({'precision': args.precision, 'model': args.model, 'rows': rows}, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
