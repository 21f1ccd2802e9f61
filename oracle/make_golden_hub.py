#!/usr/bin/env python3
"""Mint golden vectors for models built through the reference's HUB KEYWORD ARGUMENTS.   *** TEST INFRASTRUCTURE ***

The reference lets a caller override any model hyper-parameter: `create_model(experiment, **kwargs)` ends in
`config.update(kwargs)` (strhub/models/utils.py:41), and `hubconf.py:13-33` forwards its `**kwargs` there.  The two that
change tensor SHAPES on the decoder side are `charset_train` (configs/charset/36_lowercase.yaml:3, 62_mixed-case.yaml:3 —
the head is `len(charset) + 1` wide, the embedding `len(charset) + 3` rows) and `max_label_length` (`pos_queries` has
`max_label_length + 1` rows and every AR / refinement loop runs that many positions).

This script runs in the build container only.  For every entry of `oracle.synth.HUB_VARIANTS` it
  1. calls the reference's UNMODIFIED `strhub.models.utils.create_model(experiment, **kwargs)` — its YAML resolution, its
     system class, its Tokenizer — under the import stubs of oracle/make_golden_train.py (pytorch_lightning / nltk /
     timm.optim are absent here and untouched by the forward) and the timm stand-in;
  2. records the resolved configuration (`_get_config`), so that a CPU test can hold `parseq_amd.configs.get_config` to it;
  3. loads the synthetic state dict of the implied shape into `.model` with strict=True;
  4. runs the seven decode modes through the SYSTEM's `forward(images, max_length)` (system.py:87-88) and stores inputs,
     `memory`, logits, the strings and confidences of the system's own tokenizer.

Usage:  python oracle/make_golden_hub.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.make_golden import min_margin  # noqa: E402
from oracle.make_golden_train import install_stubs  # noqa: E402
from oracle.synth import HUB_VARIANTS, state_dict_fingerprint, synth_images, variant_config, variant_state_dict  # noqa: E402


def modes_for(max_label_length: int):
    """(decode_ar, refine_iters, max_length): the seven cases of make_golden.py, `max_length` scaled to the label length
    (model.py:85 takes min(max_length, max_label_length))."""
    short = min(7, max_label_length - 3)
    return {
        'nar0': (False, 0, None),
        'nar1': (False, 1, None),
        'ar0': (True, 0, None),                      # natural early exit
        'ar0_full': (True, 0, max_label_length),     # max_length given -> every step runs
        'ar0_short': (True, 0, short),               # -> L = short + 1
        'ar1': (True, 1, None),
        'ar2': (True, 2, None),
    }


@torch.inference_mode()
def run_mode(system, images, mode):
    decode_ar, refine_iters, max_length = mode
    system.model.decode_ar, system.model.refine_iters = decode_ar, refine_iters
    return system.forward(images, max_length)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--candidates', type=int, default=64)
    ap.add_argument('--keep', type=int, default=4)
    args = ap.parse_args()
    from safetensors.torch import save_file
    install_stubs()
    sys.path.insert(0, args.ref)
    from strhub.models import utils as ref_utils

    for name, (experiment, kwargs, eos_bias) in HUB_VARIANTS.items():
        cfg = variant_config(name)
        resolved = ref_utils._get_config(experiment, **kwargs)
        system = ref_utils.create_model(experiment, **kwargs).eval()
        tok = system.tokenizer
        assert len(tok) == cfg.num_tokens and (tok.eos_id, tok.bos_id, tok.pad_id) == (cfg.eos_id, cfg.bos_id, cfg.pad_id)
        assert system.model.max_label_length == cfg.max_label_length
        sd = variant_state_dict(name, seed=0)
        res = system.model.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        modes = modes_for(cfg.max_label_length)
        cand = synth_images(args.candidates, cfg, seed=4321)
        worst = torch.full((args.candidates,), float('inf'))
        for mode in ('nar0', 'ar0_full', 'ar1', 'ar2', 'nar1'):
            worst = torch.minimum(worst, min_margin(run_mode(system, cand, modes[mode])))
        # Half of the kept crops: the best-separated ones whose AR string ends strictly inside the label range (the early-exit
        # test, the refinement padding mask and the tokenizer's EOS cut all see a mixed batch); the rest: best-separated overall.
        ar_ids = run_mode(system, cand, modes['ar0_full']).argmax(-1)
        is_eos = ar_ids == tok.eos_id
        first = torch.where(is_eos.any(-1), is_eos.int().argmax(-1), torch.full((args.candidates,), ar_ids.shape[1]))
        inside = ((first >= 1) & (first < ar_ids.shape[1])).nonzero().flatten().tolist()
        inside.sort(key=lambda i: -float(worst[i]))
        chosen = [i for i in inside if float(worst[i]) > 5e-3][:args.keep // 2]
        for i in worst.argsort(descending=True).tolist():
            if len(chosen) < args.keep and i not in chosen:
                chosen.append(i)
        order = torch.tensor(sorted(chosen))
        images = cand[order].contiguous()
        out = {'images': images}
        with torch.inference_mode():
            out['memory'] = system.model.encode(images).contiguous()
        meta = {'model': name, 'experiment': experiment, 'kwargs': kwargs, 'eos_bias': eos_bias, 'resolved_config': resolved,
                'num_params': sum(p.numel() for p in system.model.parameters()), 'candidate_ids': order.tolist(),
                'sd_fingerprint': state_dict_fingerprint(sd), 'min_margin': float(worst[order].min()),
                'tokenizer': {'len': len(tok), 'eos_id': tok.eos_id, 'bos_id': tok.bos_id, 'pad_id': tok.pad_id},
                'torch': torch.__version__, 'modes': {}}
        for mode, spec in modes.items():
            logits = run_mode(system, images, spec)
            out[f'logits.{mode}'] = logits.contiguous()
            strings, probs = tok.decode(logits.softmax(-1))
            meta['modes'][mode] = {'decode_ar': spec[0], 'refine_iters': spec[1], 'max_length': spec[2], 'shape': list(logits.shape),
                                   'strings': strings, 'confidence': [float(p.prod()) for p in probs]}
        save_file(out, os.path.join(args.out, f'{name}.safetensors'))
        with open(os.path.join(args.out, f'{name}.json'), 'w') as f:
            json.dump(meta, f, indent=1)
        print(name, 'params', meta['num_params'], 'kept', order.tolist(), 'min margin', meta['min_margin'])
        for mode in modes:
            print('  ', mode, meta['modes'][mode]['shape'], meta['modes'][mode]['strings'])


def main_vitstr(ref='/root/reference', out_dir=os.path.join(ROOT, 'tests', 'golden'), candidates=32, keep=4):
    """ViTSTR built through the reference's create_model('vitstr', **kwargs): resolved configuration, logits of the system's forward at the
    full and at a shorter max_length, strings of the system's tokenizer."""
    from safetensors.torch import save_file
    from oracle import vitstr_oracle as V
    from oracle.synth import VITSTR_HUB_VARIANTS, vitstr_variant_config
    install_stubs()
    if ref not in sys.path:
        sys.path.insert(0, ref)
    from strhub.models import utils as ref_utils
    for name, kwargs in VITSTR_HUB_VARIANTS.items():
        cfg = vitstr_variant_config(name)
        resolved = ref_utils._get_config('vitstr', **kwargs)
        system = ref_utils.create_model('vitstr', **kwargs).eval()
        tok = system.tokenizer
        assert len(tok) == cfg.num_tokens and system.max_label_length == cfg.max_label_length
        sd = V.synth_state_dict(cfg, 0)
        res = system.model.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        cand = synth_images(candidates, cfg, seed=4321)
        with torch.inference_mode():
            top2 = system.forward(cand).topk(2, dim=-1).values
            margin = (top2[..., 0] - top2[..., 1]).amin(-1)
            order = margin.argsort(descending=True)[:keep].sort().values
            images = cand[order].contiguous()
            short = max(cfg.max_label_length - 3, 1)
            out = {'images': images, 'logits': system.forward(images).contiguous(), 'logits.short': system.forward(images, short).contiguous()}
        strings, probs = tok.decode(out['logits'].softmax(-1))
        meta = {'model': name, 'experiment': 'vitstr', 'kwargs': kwargs, 'resolved_config': resolved, 'short_max_length': short,
                'num_params': sum(p.numel() for p in system.model.parameters()), 'candidate_ids': order.tolist(),
                'sd_fingerprint': state_dict_fingerprint(sd), 'min_margin': float(margin[order].min()),
                'tokenizer': {'len': len(tok), 'eos_id': tok.eos_id, 'bos_id': tok.bos_id, 'pad_id': tok.pad_id},
                'shapes': {k: list(v.shape) for k, v in out.items()}, 'strings': strings, 'confidence': [float(p.prod()) for p in probs]}
        save_file(out, os.path.join(out_dir, f'{name}.safetensors'))
        with open(os.path.join(out_dir, f'{name}.json'), 'w') as f:
            json.dump(meta, f, indent=1)
        print(name, meta['num_params'], meta['shapes'], strings, meta['min_margin'])


if __name__ == '__main__':
    if '--vitstr-only' not in sys.argv:
        main()
    main_vitstr()
