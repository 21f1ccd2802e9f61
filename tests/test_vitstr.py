"""Row N4: ViTSTR — the ViT encoder with a class token and a per-token head, no decoder.

Goldens: the reference's own strhub/models/vitstr/model.py executed on the timm stand-in with synthetic weights
(oracle/make_golden_vitstr.py -> tests/golden/vitstr.*).  CPU: oracle/vitstr_oracle.py against them; host mirror of the
reference interface.  GPU: parseq_vitstr_forward through the C ABI against them (fp32 bar 1e-3, argmax- and string-identical;
bf16 against the rounding-aware oracle and the fp32 reference with the PARSeq bars).
"""
import pytest
import torch

from oracle import vitstr_oracle as V
from oracle.synth import state_dict_fingerprint

CFG = V.vitstr_config()


@pytest.fixture(scope='module')
def gold(golden):
    return golden('vitstr')


def test_synth_weights_and_param_count(gold):
    g, meta = gold
    sd = V.synth_state_dict(CFG, 0)
    assert state_dict_fingerprint(sd) == meta['sd_fingerprint']
    assert sum(v.numel() for v in sd.values()) == meta['num_params'] == 21_418_079


def test_oracle_matches_reference_vitstr(gold):
    g, meta = gold
    sd = V.synth_state_dict(CFG, 0)
    with torch.inference_mode():
        lo = V.forward(sd, CFG, g['images'])
        lo7 = V.forward(sd, CFG, g['images'], 7)
        lo1 = V.forward(sd, CFG, g['images'][:1])
    assert lo.shape == (8, 26, 95) and lo7.shape == (8, 8, 95)
    torch.testing.assert_close(lo, g['logits'], rtol=0, atol=5e-6)
    torch.testing.assert_close(lo7, g['logits.len7'], rtol=0, atol=5e-6)
    torch.testing.assert_close(lo1, g['logits.batch1'], rtol=0, atol=5e-6)
    assert torch.equal(lo.argmax(-1), g['logits'].argmax(-1))
    assert lo.shape[1] == 26 and V.forward(sd, CFG, g['images'][:1], 99).shape == (1, 26, 95)      # max_length is clamped


def test_hub_entrypoint_and_state_dict_layout():
    m = torch.hub.load('.', 'vitstr', source='local', pretrained=False)
    assert type(m).__name__ == 'ViTSTR' and m.hparams.img_size == [32, 128] and m.hparams.patch_size == [4, 8]
    spec = V.state_dict_spec(CFG)
    sd = m.model.state_dict()
    assert list(sd) == list(spec)                                  # same keys in timm's order
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    assert set(m.state_dict()) == {'model.' + k for k in spec}
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 3, 32, 128))


# ---------------------------------------------------------------------------------------------------------------- GPU
def _make(precision):
    from parseq_amd import create_model
    m = create_model('vitstr', precision=precision)
    m.model.load_state_dict(V.synth_state_dict(CFG, 0))
    return m.eval().to('cuda')


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])      # the two modes that meet the 1e-3 bar; bf16x3 is the hub default
def test_vitstr_fp32_matches_reference(gold, precision):
    from gpu_util import report
    g, meta = gold
    m = _make(precision)
    with torch.inference_mode():
        got = m(g['images'].cuda()).float().cpu()
        got7 = m(g['images'].cuda(), 7).float().cpu()
        got1 = m(g['images'][:1].cuda()).float().cpu()
    for tag, a, b in (('full', got, g['logits']), ('len7', got7, g['logits.len7']), ('batch1', got1, g['logits.batch1'])):
        d, msg = report(f'vitstr {precision} {tag}', a, b)
        assert a.shape == b.shape and d <= 1e-3, msg
        assert torch.equal(a.argmax(-1), b.argmax(-1))
    labels, _ = m.tokenizer.decode(got.softmax(-1))
    assert labels == meta['strings']
    labels2, conf = m.tokenizer.read(got.cuda())
    assert labels2 == meta['strings'] and torch.allclose(conf, torch.tensor(meta['confidence']), rtol=1e-3, atol=1e-6)
    # the inner model's forward(x, seqlen) keeps the reference's shape contract (class-token row present)
    with torch.inference_mode():
        inner = m.model(g['images'].cuda(), 27).cpu()
    assert inner.shape == (8, 27, 95) and torch.equal(inner[:, 1:], got)


@pytest.mark.gpu
def test_vitstr_bf16(gold):
    from gpu_util import report
    g, _ = gold
    m = _make('bf16')
    with torch.inference_mode():
        got = m(g['images'].cuda()).float().cpu()
        rounded = V.forward(V.synth_state_dict(CFG, 0), CFG, g['images'], rounding='bf16')
    d_r, msg_r = report('vitstr bf16 vs rounding oracle', got, rounded)
    d_f, msg_f = report('vitstr bf16 vs fp32 reference', got, g['logits'])
    assert d_r <= 3e-2, msg_r
    assert d_f <= 6e-2, msg_f


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 5, 64])
def test_vitstr_batch_sizes_bf16(gold, batch):
    g, _ = gold
    m = _make('bf16')
    idx = torch.arange(batch) % 8
    with torch.inference_mode():
        small = m(g['images'].cuda()).float().cpu()
        got = m(g['images'][idx].cuda()).float().cpu()
    assert (got - small[idx]).abs().max().item() <= 1e-5


@pytest.mark.gpu
def test_vitstr_plan_refuses_parseq_entry_points(gold):
    g, _ = gold
    m = _make('fp32')
    from parseq_amd import _native
    import ctypes as C
    plan = m.model._plan(8)
    out = torch.empty(8, 26, 95, device='cuda')
    n = C.c_int(0)
    status = _native.lib().parseq_forward(plan, _native.ptr(g['images'].cuda()), 0, 8, 1, 1, 26, _native.ptr(out), C.byref(n), _native.stream_ptr())
    assert status != 0 and b'ViTSTR' in _native.lib().parseq_last_error()


@pytest.mark.gpu
def test_vitstr_batch_512_tail_rows_take_the_generic_kernels(gold):
    """512 x 129 rows = 516 row tiles of 128: the 512 leading tiles go to the fused kernels (whole rounds on 256 CUs), the last
    512 rows to the per-op kernels (lib_encode.hip: main_rows).  Images fully inside the leading tiles reproduce the small-batch
    result bit for bit; the last four (different fp32 summation order in two GEMMs per layer) within the bf16 bar."""
    g, _ = gold
    m = _make('bf16')
    idx = torch.arange(512) % 8
    with torch.inference_mode():
        small = m(g['images'].cuda()).float().cpu()
        got = m(g['images'][idx].cuda()).float().cpu()
    d = (got - small[idx]).abs().amax(dim=(1, 2))
    assert d[:508].max().item() <= 1e-5
    assert d[508:].max().item() <= 3e-2
