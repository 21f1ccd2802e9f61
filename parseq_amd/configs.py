"""Hyper-parameters of the PARSeq experiments, restated as data.

The reference resolves these from YAML with a hand-rolled Hydra emulation (`strhub/models/utils.py:25-44`):
`configs/main.yaml['model']` <- `configs/charset/94_full.yaml` <- `configs/model/parseq.yaml` <-
`configs/experiment/<name>.yaml['model']` <- keyword overrides, then `lr` cast to float.  The merged result for the three
PARSeq experiments is written out here (values are facts about the published models; file:line cited per block).
"""
from __future__ import annotations

import copy

# configs/charset/94_full.yaml:3
CHARSET_94_FULL = ("0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
                   "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")

_BASE = {
    # configs/main.yaml:7-17 (model section)
    '_convert_': 'all',
    'img_size': [32, 128],
    'max_label_length': 25,
    'charset_train': CHARSET_94_FULL,
    'charset_test': '0123456789abcdefghijklmnopqrstuvwxyz',
    'batch_size': 384,
    'weight_decay': 0.0,
    'warmup_pct': 0.075,
    # configs/model/parseq.yaml:1-25
    'name': 'parseq',
    '_target_': 'strhub.models.parseq.system.PARSeq',
    'patch_size': [4, 8],
    'embed_dim': 384,
    'enc_num_heads': 6,
    'enc_mlp_ratio': 4,
    'enc_depth': 12,
    'dec_num_heads': 12,
    'dec_mlp_ratio': 4,
    'dec_depth': 1,
    'lr': 7e-4,
    'perm_num': 6,
    'perm_forward': True,
    'perm_mirrored': True,
    'dropout': 0.1,
    'decode_ar': True,
    'refine_iters': 1,
}

EXPERIMENTS = {
    'parseq': {},                                                        # configs/experiment/parseq.yaml (no overrides)
    'parseq-tiny': {'name': 'parseq-tiny', 'embed_dim': 192,             # configs/experiment/parseq-tiny.yaml:5-9
                    'enc_num_heads': 3, 'dec_num_heads': 6},
    'parseq-patch16-224': {'img_size': [224, 224], 'patch_size': [16, 16]},  # configs/experiment/parseq-patch16-224.yaml:5-7
}


# configs/model/vitstr.yaml:1-14 over configs/main.yaml:7-17 and the 94-char charset; configs/experiment/vitstr.yaml:5-7
# overrides the model's 224 x 224 / 16 x 16 defaults with the 32 x 128 crops and 4 x 8 patches every released model uses
_VITSTR = {
    '_convert_': 'all', 'img_size': [32, 128], 'max_label_length': 25, 'charset_train': CHARSET_94_FULL,
    'charset_test': '0123456789abcdefghijklmnopqrstuvwxyz', 'batch_size': 384, 'weight_decay': 0.0, 'warmup_pct': 0.075,
    'name': 'vitstr', '_target_': 'strhub.models.vitstr.system.ViTSTR', 'patch_size': [4, 8], 'embed_dim': 384, 'num_heads': 6,
    'lr': 8.9e-4,
}


def get_config(experiment: str, **kwargs) -> dict:
    if experiment == 'vitstr':
        config = copy.deepcopy(_VITSTR)
        config.update(kwargs)
        config['lr'] = float(config['lr'])
        return config
    if experiment not in EXPERIMENTS:
        raise FileNotFoundError(experiment)
    config = copy.deepcopy(_BASE)
    config.update(EXPERIMENTS[experiment])
    config.update(kwargs)
    config['lr'] = float(config['lr'])
    return config
