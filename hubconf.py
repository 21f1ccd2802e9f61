"""torch.hub entrypoints: the names, positional order and defaults of the reference's `hubconf.py:6-58` for the model
families this repository implements (PARSeq small / tiny / patch16-224, ViTSTR).

    torch.hub.load('<repo dir>', 'parseq', source='local', pretrained=False, decode_ar=True, refine_iters=1)

`dependencies` names torch alone: `torch.hub` refuses a hubconf whose dependencies cannot be imported, and this backend
needs neither pytorch_lightning nor timm.  One keyword goes beyond the reference: `precision='bf16x3' | 'bf16' | 'fp32'` picks the
arithmetic mode of the HIP library; the default, 'bf16x3', is the one whose logits equal the reference's fp32
CPU path within 1e-3 at matrix-core speed — 'bf16' is the faster throughput mode (3e-2 on the logits).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from parseq_amd.utils import create_model  # noqa: E402

dependencies = ['torch']


def _parseq_entry(experiment: str, summary: str):
    def entry(pretrained: bool = False, decode_ar: bool = True, refine_iters: int = 1, **kwargs):
        return create_model(experiment, pretrained, decode_ar=decode_ar, refine_iters=refine_iters, **kwargs)

    entry.__doc__ = (f'{summary}\n\n'
                     'pretrained   -- download and load the released checkpoint of this experiment\n'
                     'decode_ar    -- autoregressive decoding (True) or one non-autoregressive pass (False)\n'
                     'refine_iters -- number of cloze refinement passes after decoding\n'
                     'precision    -- "bf16x3" (default: within 1e-3 of the reference), "bf16" (throughput) or "fp32": arithmetic mode of libparseq_hip\n'
                     'Any other keyword overrides the experiment configuration, as in the reference.')
    return entry


parseq_tiny = _parseq_entry('parseq-tiny', 'PARSeq-Ti: 32x128 crops, 4x8 patches, width 192 (6.0 M parameters).')
parseq = _parseq_entry('parseq', 'PARSeq-S: 32x128 crops, 4x8 patches, width 384 (23.8 M parameters).')
parseq_patch16_224 = _parseq_entry('parseq-patch16-224', 'PARSeq-S on 224x224 crops with 16x16 patches (196 visual tokens).')
for _name in ('parseq_tiny', 'parseq', 'parseq_patch16_224'):
    globals()[_name].__name__ = globals()[_name].__qualname__ = _name


def vitstr(pretrained: bool = False, **kwargs):
    """ViTSTR-S: ViT encoder with a class token and a per-token head; 32x128 crops, 4x8 patches, width 384.

    pretrained -- download and load the released checkpoint; other keywords override the experiment configuration."""
    return create_model('vitstr', pretrained, **kwargs)
