"""torch.hub entrypoints with the reference's names and signatures (reference `hubconf.py:6-33`).

    torch.hub.load('<repo dir>', 'parseq', source='local', pretrained=False, decode_ar=True, refine_iters=1)

Only the PARSeq family is provided (the path this repository accelerates); `dependencies` lists just torch, because
`torch.hub` refuses to load a hubconf whose dependencies are not importable and this backend needs neither
pytorch_lightning nor timm.  Extra keyword `precision='bf16'|'fp32'` selects the arithmetic mode of the HIP library.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from parseq_amd.utils import create_model  # noqa: E402

dependencies = ['torch']


def parseq_tiny(pretrained: bool = False, decode_ar: bool = True, refine_iters: int = 1, **kwargs):
    """
    PARSeq tiny model (img_size=128x32, patch_size=8x4, d_model=192)
    @param pretrained: (bool) Use pretrained weights
    @param decode_ar: (bool) use AR decoding
    @param refine_iters: (int) number of refinement iterations to use
    """
    return create_model('parseq-tiny', pretrained, decode_ar=decode_ar, refine_iters=refine_iters, **kwargs)


def parseq(pretrained: bool = False, decode_ar: bool = True, refine_iters: int = 1, **kwargs):
    """
    PARSeq base model (img_size=128x32, patch_size=8x4, d_model=384)
    @param pretrained: (bool) Use pretrained weights
    @param decode_ar: (bool) use AR decoding
    @param refine_iters: (int) number of refinement iterations to use
    """
    return create_model('parseq', pretrained, decode_ar=decode_ar, refine_iters=refine_iters, **kwargs)


def parseq_patch16_224(pretrained: bool = False, decode_ar: bool = True, refine_iters: int = 1, **kwargs):
    """
    PARSeq base model (img_size=224x224, patch_size=16x16, d_model=384)
    Constructible (parameters, state_dict); its 196-token encoder is not yet covered by the gfx950 attention kernel, so
    forward() raises until row N4 of SURVEY.md section 8f lands.
    """
    return create_model('parseq-patch16-224', pretrained, decode_ar=decode_ar, refine_iters=refine_iters, **kwargs)


def vitstr(pretrained: bool = False, **kwargs):
    """
    ViTSTR small model (img_size=32x128, patch_size=4x8, d_model=384)
    @param pretrained: (bool) Use pretrained weights
    """
    return create_model('vitstr', pretrained, **kwargs)
