"""Mint the ViTSTR fixtures (row N4): runs the REFERENCE's own strhub/models/vitstr/model.py (class ViTSTR) on the timm
stand-in with the synthetic weights, exactly as strhub/models/vitstr/system.py:51-60,76-82 builds and calls it.
    python oracle/make_golden_vitstr.py   ->  tests/golden/vitstr.{safetensors,json}
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import timm_standin  # noqa: E402
from oracle import vitstr_oracle as V  # noqa: E402
from oracle.synth import state_dict_fingerprint, synth_images  # noqa: E402

CHARSET_94 = ("0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
              "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")


def main(ref_root='/root/reference', out_dir=os.path.join(ROOT, 'tests', 'golden'), candidates=64, keep=8):
    from safetensors.torch import save_file
    timm_standin.install()
    sys.path.insert(0, ref_root)
    from strhub.data.utils import Tokenizer
    from strhub.models.vitstr.model import ViTSTR
    cfg = V.vitstr_config()
    tok = Tokenizer(CHARSET_94)
    sd = V.synth_state_dict(cfg, 0)
    model = ViTSTR(img_size=list(cfg.img_size), patch_size=list(cfg.patch_size), depth=12, mlp_ratio=4, qkv_bias=True,
                   embed_dim=cfg.embed_dim, num_heads=cfg.enc_num_heads, num_classes=len(tok) - 2).eval()
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys

    @torch.inference_mode()
    def system_forward(images, max_length=None):          # vitstr/system.py:76-82
        max_length = cfg.max_label_length if max_length is None else min(max_length, cfg.max_label_length)
        return model.forward(images, max_length + 2)[:, 1:]

    cand = synth_images(candidates, cfg, seed=4321)
    top2 = system_forward(cand).topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).amin(-1)
    order = margin.argsort(descending=True)[:keep].sort().values
    images = cand[order].contiguous()
    out = {'images': images, 'logits': system_forward(images).contiguous(), 'logits.len7': system_forward(images, 7).contiguous(),
           'logits.batch1': system_forward(images[:1]).contiguous()}
    with torch.inference_mode():
        out['features'] = model.forward_features(images).contiguous()
    strings, probs = tok.decode(out['logits'].softmax(-1))
    meta = {'model': 'vitstr', 'num_params': sum(p.numel() for p in model.parameters()), 'candidate_ids': order.tolist(),
            'sd_fingerprint': state_dict_fingerprint(sd), 'min_margin': float(margin[order].min()), 'torch': torch.__version__,
            'shapes': {k: list(v.shape) for k, v in out.items()}, 'strings': strings, 'confidence': [float(p.prod()) for p in probs]}
    save_file(out, os.path.join(out_dir, 'vitstr.safetensors'))
    with open(os.path.join(out_dir, 'vitstr.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print(meta['num_params'], meta['shapes'], meta['strings'][:4], meta['min_margin'])


if __name__ == '__main__':
    main()
