"""Row N4: parseq-patch16-224 — 224 x 224 crops, 16 x 16 patches, 196 visual tokens — through the same
library.  Shares every kernel with the 32 x 128 models except encoder attention and decoder cross-attention, which take the
token-count-generic kernels.  Goldens from the reference's own model code (tests/golden/parseq-patch16-224.*)."""
import pytest
import torch

from gpu_util import DEV, make_model, report
from oracle import parseq_oracle as O
from oracle.synth import CONFIGS, synth_state_dict

pytestmark = pytest.mark.gpu
NAME = 'parseq-patch16-224'
MODES = {'nar0': (False, 0, None), 'nar1': (False, 1, None), 'ar0': (True, 0, None), 'ar0_full': (True, 0, 25),
         'ar0_len7': (True, 0, 7), 'ar1': (True, 1, None), 'ar2': (True, 2, None)}


@pytest.fixture(scope='module')
def models():
    return {p: make_model(NAME, p) for p in ('fp32', 'bf16', 'bf16x3')}


def _run(m, images, mode):
    ar, ri, ml = MODES[mode]
    m.model.decode_ar, m.model.refine_iters = ar, ri
    with torch.inference_mode():
        out = m(images, ml)
    torch.cuda.synchronize()
    return out.float().cpu()


def test_encoder_memory_fp32(models, golden):
    g, _ = golden(NAME)
    with torch.inference_mode():
        mem = models['fp32'].model.encode(g['images'].to(DEV)).cpu()
    d, msg = report('patch16 memory fp32', mem, g['memory'])
    assert mem.shape == (2, 196, 384) and d <= 1e-4, msg


@pytest.mark.parametrize('mode', list(MODES))
def test_forward_fp32_matches_reference(models, golden, mode):
    g, meta = golden(NAME)
    got = _run(models['fp32'], g['images'].to(DEV), mode)
    want = g[f'logits.{mode}']
    assert list(got.shape) == meta['modes'][mode]['shape']
    d, msg = report(f'patch16 {mode} fp32', got, want)
    assert d <= 1e-3, msg
    assert torch.equal(got.argmax(-1), want.argmax(-1))
    labels, _ = models['fp32'].tokenizer.decode(got.softmax(-1))
    assert labels == meta['modes'][mode]['strings']


@pytest.mark.parametrize('mode', ['nar0', 'ar0', 'ar1'])
def test_forward_bf16x3_matches_reference(models, golden, mode):
    """The exact-tolerance mode on 196 tokens: encoder attention through attn_split_n_kernel (encoder_attn.h: the token-count-generic
    MFMA kernel in the three-product arithmetic; round 5 — it was the scalar generic kernel), everything else the split GEMMs.  Same
    bar as fp32: 1e-3 on the logits, identical arg-max and strings.  PARSEQ_ATTN_GENERIC=1 keeps the scalar kernel (the A/B)."""
    g, meta = golden(NAME)
    got = _run(models['bf16x3'], g['images'].to(DEV), mode)
    want = g[f'logits.{mode}']
    d, msg = report(f'patch16 {mode} bf16x3', got, want)
    assert d <= 1e-3, msg
    assert torch.equal(got.argmax(-1), want.argmax(-1))
    labels, _ = models['bf16x3'].tokenizer.decode(got.softmax(-1))
    assert labels == meta['modes'][mode]['strings']


@pytest.mark.parametrize('mode', ['nar0', 'ar1', 'ar2'])
def test_forward_bf16(models, golden, mode):
    g, _ = golden(NAME)
    got = _run(models['bf16'], g['images'].to(DEV), mode)
    want = g[f'logits.{mode}']
    ar, ri, ml = MODES[mode]
    with torch.inference_mode():
        rounded = O.forward(synth_state_dict(CONFIGS[NAME], 0), CONFIGS[NAME], g['images'], ml, decode_ar=ar, refine_iters=ri,
                            rounding='bf16')
    d_r, msg_r = report(f'patch16 {mode} bf16 vs rounding oracle', got, rounded)
    d_f, msg_f = report(f'patch16 {mode} bf16 vs fp32 reference', got, want)
    assert d_r <= 3e-2, msg_r
    assert d_f <= 6e-2, msg_f


def test_batch_sizes_and_row_tails_bf16(models, golden):
    """196-token images make every 128-row tile straddle images and leave ragged tails (B = 1: 196 rows; B = 5: 980)."""
    g, _ = golden(NAME)
    m = models['bf16']
    small = _run(m, g['images'].to(DEV), 'ar1')
    for batch in (1, 5, 9):
        idx = torch.arange(batch) % 2
        got = _run(m, g['images'][idx].to(DEV), 'ar1')
        d = (got - small[idx]).abs().max().item()
        print(f'[patch16 batch {batch}] max|d| vs batch-2 run {d:.3e}')
        assert d <= 1e-5


def test_hub_entrypoint():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = torch.hub.load(root, 'parseq_patch16_224', source='local', pretrained=False).eval().to(DEV)
    with torch.inference_mode():
        out = m(torch.zeros(1, 3, 224, 224, device=DEV))
    assert out.shape[0] == 1 and out.shape[2] == 95
