#!/usr/bin/env python3
"""Mint golden vectors by executing the REFERENCE's own model code.   *** TEST INFRASTRUCTURE ***

Runs only in the build container (needs /root/reference, which does not exist on the GPU box).  It
  1. installs the timm stand-in (oracle/timm_standin.py) — timm is an un-vendored dependency of the reference;
  2. imports the reference's unmodified `strhub.models.parseq.model.PARSeq` (model.py + modules.py) and
     `strhub.data.utils.Tokenizer` from /root/reference;
  3. loads the synthetic state_dict (oracle/synth.py) with strict=True — which also proves the key set / shapes
     of SURVEY.md section 8(b);
  4. selects crops whose decisions are well separated (min top1-top2 logit margin), so that fp32-level
     reordering noise cannot flip an argmax, and runs every decode mode;
  5. writes tests/golden/<model>.safetensors (+ .json with strings, lengths, checksums).

Nothing from /root/reference is copied: only tensors it computed are stored.

Usage:  python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import timm_standin  # noqa: E402
from oracle.synth import CONFIGS, state_dict_fingerprint, synth_images, synth_state_dict  # noqa: E402

CHARSET_94 = ("0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
              "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")   # configs/charset/94_full.yaml:3 (a fact, restated)

MODES = {
    # name: (decode_ar, refine_iters, max_length)
    'nar0': (False, 0, None),
    'nar1': (False, 1, None),
    'ar0': (True, 0, None),        # natural early exit -> L <= 26
    'ar0_full': (True, 0, 25),     # max_length given -> always 26 steps
    'ar0_len7': (True, 0, 7),      # max_length=7 -> L = 8
    'ar1': (True, 1, None),
    'ar2': (True, 2, None),
}


def build_reference(ref_root: str, cfg, sd):
    timm_standin.install()
    sys.path.insert(0, ref_root)
    from strhub.data.utils import Tokenizer
    from strhub.models.parseq.model import PARSeq
    tok = Tokenizer(CHARSET_94)
    assert len(tok) == cfg.num_tokens and (tok.eos_id, tok.bos_id, tok.pad_id) == (cfg.eos_id, cfg.bos_id, cfg.pad_id)
    model = PARSeq(len(tok), cfg.max_label_length, list(cfg.img_size), list(cfg.patch_size), cfg.embed_dim,
                   cfg.enc_num_heads, cfg.enc_mlp_ratio, cfg.enc_depth, cfg.dec_num_heads, cfg.dec_mlp_ratio,
                   cfg.dec_depth, decode_ar=True, refine_iters=1, dropout=0.1).eval()
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model, tok


@torch.inference_mode()
def run_mode(model, tok, images, mode):
    decode_ar, refine_iters, max_length = MODES[mode]
    model.decode_ar, model.refine_iters = decode_ar, refine_iters
    return model.forward(tok, images, max_length)


def min_margin(logits: torch.Tensor) -> torch.Tensor:
    top2 = logits.topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]).amin(dim=-1)   # per image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--candidates', type=int, default=96)
    ap.add_argument('--keep', type=int, default=8)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--only', default=None, help='mint one configuration only')
    args = ap.parse_args()
    from safetensors.torch import save_file
    torch.manual_seed(0)
    os.makedirs(args.out, exist_ok=True)

    for name, cfg in CONFIGS.items():
        if args.only and name != args.only:
            continue
        # the 224 x 224 configuration keeps 2 of 64 candidates: its crops are 37x larger and it shares every kernel but
        # the attention ones with the 32 x 128 models
        n_cand, n_keep = (64, 2) if cfg.num_patches != 128 else (args.candidates, args.keep)
        sd = synth_state_dict(cfg, seed=args.seed)
        model, tok = build_reference(args.ref, cfg, sd)
        n_params = sum(p.numel() for p in model.parameters())
        cand = synth_images(n_cand, cfg, seed=1234)
        # rank candidates by the worst decision margin over every mode
        worst = torch.full((n_cand,), float('inf'))
        for mode in ('nar0', 'ar0_full', 'ar1', 'ar2', 'nar1'):
            worst = torch.minimum(worst, min_margin(run_mode(model, tok, cand, mode)))
        order = worst.argsort(descending=True)[:n_keep].sort().values
        images = cand[order].contiguous()
        out = {'images': images}
        meta = {'model': name, 'seed': args.seed, 'num_params': n_params, 'candidate_ids': order.tolist(),
                'sd_fingerprint': state_dict_fingerprint(sd), 'min_margin': float(worst[order].min()),
                'torch': torch.__version__, 'modes': {}}
        with torch.inference_mode():
            out['memory'] = model.encode(images).contiguous()
        for mode in MODES:
            logits = run_mode(model, tok, images, mode)
            out[f'logits.{mode}'] = logits.contiguous()
            strings, probs = tok.decode(logits.softmax(-1))
            meta['modes'][mode] = {'shape': list(logits.shape), 'strings': strings,
                                   'confidence': [float(p.prod()) for p in probs]}
        # batch-1 run of image 0 through the default mode (batch invariance + p_i.squeeze() edge case)
        out['logits.ar1.batch1'] = run_mode(model, tok, images[:1], 'ar1').contiguous()
        save_file(out, os.path.join(args.out, f'{name}.safetensors'))
        with open(os.path.join(args.out, f'{name}.json'), 'w') as f:
            json.dump(meta, f, indent=1)
        print(name, 'params', n_params, 'kept', order.tolist(), 'min margin', meta['min_margin'])
        for mode in MODES:
            print('  ', mode, meta['modes'][mode]['shape'], meta['modes'][mode]['strings'][:4])


if __name__ == '__main__':
    main()
