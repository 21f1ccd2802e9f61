#!/usr/bin/env python3
"""Mint golden vectors for row N3 (training step) by executing the REFERENCE's own system.py.   *** TEST INFRASTRUCTURE ***

Runs only in the build container (needs /root/reference).  The reference's `strhub.models.parseq.system` imports three
packages that are absent here — pytorch_lightning, nltk, timm.optim — for things the training step's arithmetic never
touches (the LightningModule base class, `edit_distance` in the eval step, the optimiser factory).  They are replaced
by the minimal stubs below; `gen_tgt_perms`, `generate_attn_masks` and `training_step` then run unmodified.

Writes
  tests/golden/perms.json               `gen_tgt_perms` for every label length 1..25 and every (perm_num, perm_forward,
                                        perm_mirrored) the reference's configs use, under fixed numpy / torch seeds
  tests/golden/parseq_train.safetensors one `training_step` of PARSeq-S (synthetic weights of oracle/synth.py, dropout off
  tests/golden/parseq_train.json        via .eval()): the permutations drawn, the loss, and its gradient w.r.t. every
                                        parameter (full tensors for the small ones, L2 norm + a checksum for all)

Usage:  python oracle/make_golden_train.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import timm_standin  # noqa: E402
from oracle.make_golden import CHARSET_94  # noqa: E402
from oracle.synth import CONFIGS, synth_images, synth_state_dict  # noqa: E402

LABELS = ['Hello', 'a', 'MI355X', 'parallel-decoding', 'x7', 'Permuted_AR_Sequence(25)!', 'stop', 'W0rld#42']
PERM_SETTINGS = [(6, True, True), (1, True, False), (2, True, True), (6, False, True), (5, True, False), (12, True, True)]
# gradients small enough to store whole (the rest are pinned by norm + checksum)
FULL_GRADS = ['head.bias', 'head.weight', 'pos_queries', 'decoder.norm.weight', 'decoder.norm.bias',
              'decoder.layers.0.norm_q.weight', 'decoder.layers.0.norm_c.bias', 'decoder.layers.0.self_attn.in_proj_bias',
              'decoder.layers.0.cross_attn.out_proj.bias', 'text_embed.embedding.weight', 'encoder.pos_embed',
              'encoder.norm.weight', 'encoder.blocks.0.attn.qkv.bias', 'encoder.blocks.11.mlp.fc2.bias',
              'encoder.patch_embed.proj.bias']


def install_stubs():
    timm_standin.install()
    timm = sys.modules['timm']
    optim = types.ModuleType('timm.optim')
    optim.create_optimizer_v2 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('optimizer stub'))
    timm.optim = optim
    sys.modules['timm.optim'] = optim

    pl = types.ModuleType('pytorch_lightning')

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self._device = torch.device('cpu')
            self.logged = {}

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, value, *a, **k):
            self.logged[name] = value

    pl.LightningModule = LightningModule
    util = types.ModuleType('pytorch_lightning.utilities')
    tps = types.ModuleType('pytorch_lightning.utilities.types')
    tps.STEP_OUTPUT = object
    pl.utilities, util.types = util, tps
    nltk = types.ModuleType('nltk')
    nltk.edit_distance = lambda a, b: (_ for _ in ()).throw(RuntimeError('nltk stub'))
    sys.modules.update({'pytorch_lightning': pl, 'pytorch_lightning.utilities': util,
                        'pytorch_lightning.utilities.types': tps, 'nltk': nltk})


def build_system(ref_root, cfg, perm_num=6, perm_forward=True, perm_mirrored=True):
    install_stubs()
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    from strhub.models.parseq.system import PARSeq
    return PARSeq(CHARSET_94, CHARSET_94, cfg.max_label_length, 384, 7e-4, 0.075, 0.0, list(cfg.img_size), list(cfg.patch_size),
                  cfg.embed_dim, cfg.enc_num_heads, cfg.enc_mlp_ratio, cfg.enc_depth, cfg.dec_num_heads, cfg.dec_mlp_ratio,
                  cfg.dec_depth, perm_num, perm_forward, perm_mirrored, True, 1, 0.1)


def checksum(t: torch.Tensor) -> float:
    """Position-weighted sum: sensitive to transposition, cheap to restate in a test."""
    f = t.detach().double().flatten()
    w = torch.arange(1, f.numel() + 1, dtype=torch.float64) % 97 + 1
    return float((f * w).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    args = ap.parse_args()
    from safetensors.torch import save_file
    cfg = CONFIGS['parseq']

    # ---- permutation sampler ---------------------------------------------------------------------------
    perm_cases = []
    for perm_num, fwd, mir in PERM_SETTINGS:
        system = build_system(args.ref, cfg, perm_num, fwd, mir)
        for T in range(1, cfg.max_label_length + 1):
            np_seed, torch_seed = 1000 + T, 2000 + T
            system.rng = np.random.default_rng(np_seed)
            torch.manual_seed(torch_seed)
            try:
                perms = system.gen_tgt_perms(torch.zeros(3, T + 2, dtype=torch.long))
            except RuntimeError as e:
                # system.py:122 stacks an empty list when perm_forward is off and the pool branch (< 5 chars) is taken
                assert not fwd and 1 < T < 5, e
                continue
            perm_cases.append({'perm_num': perm_num, 'perm_forward': fwd, 'perm_mirrored': mir, 'num_chars': T,
                               'np_seed': np_seed, 'torch_seed': torch_seed, 'perms': perms.tolist()})
    # attention masks of a few permutations (bool -> 0/1 lists)
    mask_cases = []
    system = build_system(args.ref, cfg)
    for case in perm_cases[4:25:5]:
        for perm in case['perms'][:3]:
            cm, qm = system.generate_attn_masks(torch.tensor(perm))
            mask_cases.append({'perm': perm, 'content_mask': cm.int().tolist(), 'query_mask': qm.int().tolist()})
    with open(os.path.join(args.out, 'perms.json'), 'w') as f:
        json.dump({'torch': torch.__version__, 'numpy': np.__version__, 'perms': perm_cases, 'masks': mask_cases}, f)
    print('perm cases', len(perm_cases), 'mask cases', len(mask_cases))

    # ---- one training step ------------------------------------------------------------------------------
    sd = synth_state_dict(cfg, seed=0)
    system = build_system(args.ref, cfg).eval()          # .eval(): dropout off, the only non-reproducible part of the step
    system.model.load_state_dict(sd, strict=True)
    images = synth_images(len(LABELS), cfg, seed=4321)
    system.rng = np.random.default_rng(11)
    torch.manual_seed(22)
    drawn = []
    gen = system.gen_tgt_perms
    system.gen_tgt_perms = lambda tgt: drawn.append(gen(tgt)) or drawn[-1]
    loss = system.training_step((images, LABELS), 0)
    loss.backward()
    out = {'images': images, 'perms': drawn[0].to(torch.int32), 'loss': loss.detach().reshape(1)}
    grads = {}
    for k, p in system.model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        grads[k] = {'norm': float(g.double().norm()), 'checksum': checksum(g), 'none': p.grad is None}
        if k in FULL_GRADS:
            out['grad.' + k] = g.detach().contiguous()
    save_file(out, os.path.join(args.out, 'parseq_train.safetensors'))
    with open(os.path.join(args.out, 'parseq_train.json'), 'w') as f:
        json.dump({'labels': LABELS, 'loss': float(loss.detach()), 'np_seed': 11, 'torch_seed': 22, 'image_seed': 4321,
                   'perms': drawn[0].tolist(), 'grads': grads, 'torch': torch.__version__}, f, indent=1)
    print('loss', float(loss.detach()), 'perms', drawn[0].shape, 'grad keys', len(grads),
          'total grad norm', sum(v['norm'] ** 2 for v in grads.values()) ** 0.5)


if __name__ == '__main__':
    main()
