"""The small-batch route of the bf16x3 mode (round 6; lib_internal.h `small_batch_max`, lib_encode.hip): up to 64 crops the PARSeq-S
encoder runs as per-operation launches (their tiles spread an image's rows and columns over the device) instead of the one-launch kernel
(one workgroup per image: a batch of B images keeps B of 256 compute units busy for 3.3 ms).  Same arithmetic, another summation order.

These tests run with the library's DEFAULT settings (every other GPU test pins the route off so that the goldens keep reaching the
one-launch kernels: tests/conftest.py) and hold the route to the reference-minted goldens, to the one-launch route, and to its boundary.
"""
import pytest
import torch

from gpu_util import DEV, make_model, report
from oracle.synth import CONFIGS, HUB_VARIANTS, synth_images, variant_state_dict

pytestmark = [pytest.mark.gpu, pytest.mark.small_batch_route]

MODES = {'nar0': (False, 0, None), 'nar1': (False, 1, None), 'ar0': (True, 0, None), 'ar0_full': (True, 0, 25),
         'ar0_len7': (True, 0, 7), 'ar1': (True, 1, None), 'ar2': (True, 2, None)}


def _run(m, images, mode, **kw):
    ar, ri, ml = MODES[mode]
    m.model.decode_ar, m.model.refine_iters = ar, ri
    with torch.inference_mode():
        out = m(images, ml, **kw)
    torch.cuda.synchronize()
    return out.float().cpu()


@pytest.fixture(scope='module')
def default_model():
    return make_model('parseq', 'bf16x3')


@pytest.mark.parametrize('mode', list(MODES))
def test_default_route_matches_reference(default_model, golden, mode):
    """The goldens (8 crops: the small-batch route) — 1e-3, arg-max, strings, default call and an explicit slot."""
    g, meta = golden('parseq')
    for kw in ({}, {'slot': 1}):
        got = _run(default_model, g['images'].to(DEV), mode, **kw)
        ref = g[f'logits.{mode}']
        assert list(got.shape) == list(ref.shape)
        err, msg = report(f'small-batch route {mode} {kw}', got, ref)
        assert err <= 1e-3, msg
        assert torch.equal(got.argmax(-1), ref.argmax(-1)), msg
        strings, _ = default_model.tokenizer.decode(got.softmax(-1))
        assert strings == meta['modes'][mode]['strings']


def test_default_route_memory_and_decode_idiom(default_model, golden):
    g, _ = golden('parseq')
    mem = default_model.model.encode(g['images'].to(DEV))
    err, msg = report('small-batch route memory', mem.cpu(), g['memory'])
    assert err <= 5e-4, msg


def test_routes_agree_and_the_boundary_is_where_it_says(default_model, monkeypatch):
    """Batch 64 (per-operation launches) and batch 65 (one launch) against the one-launch route forced at both sizes: <= 1e-4 on every
    logit at 64 (another summation order), bit-identical at 65 (the same kernel); the crops are distinct random crops, AR + 1 refinement
    and the feedback-free pass."""
    x = synth_images(65, CONFIGS['parseq'], seed=21).to(DEV)
    monkeypatch.setenv('PARSEQ_SMALL_BATCH', '0')
    forced = make_model('parseq', 'bf16x3')
    one = {(mode, B): _run(forced, x[:B], mode) for mode in ('ar1', 'nar0') for B in (64, 65)}
    monkeypatch.delenv('PARSEQ_SMALL_BATCH')
    for mode in ('ar1', 'nar0'):
        d64 = (_run(default_model, x[:64], mode) - one[(mode, 64)]).abs()
        same = (_run(default_model, x[:64], mode).argmax(-1) == one[(mode, 64)].argmax(-1)).all(-1)
        print(f'[routes {mode}] batch 64: rows with identical decisions {int(same.sum())}/64, max|d| on them {d64[same].max().item():.3e}')
        assert same.float().mean() >= 0.95 and 0 < d64[same].max().item() <= 1e-4        # > 0: the other route really ran
        assert torch.equal(_run(default_model, x[:65], mode), one[(mode, 65)])
    monkeypatch.setenv('PARSEQ_SMALL_BATCH', '8')                                          # the threshold is the switch's value
    moved = make_model('parseq', 'bf16x3')
    assert torch.equal(_run(moved, x[:9], 'nar0'), _run(forced, x[:9], 'nar0'))            # 9 > 8: the one-launch kernel (`forced` keeps its plans)
    assert not torch.equal(_run(moved, x[:8], 'nar0'), _run(forced, x[:8], 'nar0'))        # 8: per-operation launches


@pytest.mark.parametrize('variant', ['parseq_c36_len10', 'parseq_c62'])
def test_default_route_on_hub_kwarg_models(variant, golden):
    from parseq_amd import create_model
    g, meta = golden(variant)
    experiment, kwargs, _ = HUB_VARIANTS[variant]
    m = create_model(experiment, precision='bf16x3', **kwargs)
    m.model.load_state_dict(variant_state_dict(variant))
    m = m.eval().to(DEV)
    for mode in ('ar1', 'ar0', 'nar1', 'ar2'):
        spec = meta['modes'][mode]
        m.model.decode_ar, m.model.refine_iters = spec['decode_ar'], spec['refine_iters']
        with torch.inference_mode():
            got = m(g['images'].to(DEV), spec['max_length']).float().cpu()
        ref = g[f'logits.{mode}']
        err, msg = report(f'small-batch route {variant} {mode}', got, ref)
        assert err <= 1e-3 and torch.equal(got.argmax(-1), ref.argmax(-1)), msg
        strings, _ = m.tokenizer.decode(got.softmax(-1))
        assert strings == spec['strings']


def test_uint8_and_bf16_inputs_on_the_default_route(default_model):
    """The per-operation patch embedding normalises raw u8 crops exactly like the one-launch head does (ToTensor + Normalize(0.5, 0.5))."""
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (5, 3, 32, 128), generator=g, dtype=torch.uint8)
    f32 = (u8.float() / 255 - 0.5) / 0.5
    a = _run(default_model, u8.to(DEV), 'ar1')
    b = _run(default_model, f32.to(DEV), 'ar1')
    assert torch.equal(a, b)


def test_routes_alternate_on_one_plan(default_model, golden):
    """One plan (workspace capacity 128) serves both routes in turn: 65 crops (one launch: K / V of the memory as 24-bit rows), the 8 golden
    crops (per-operation launches: f32 rows), 65 again — the plan must remember which format its K / V hold (parseq_plan::kv24) and nothing
    of one route's state may leak into the other: every result equals its first occurrence bit for bit, and the goldens hold in between."""
    g, _ = golden('parseq')
    x65 = synth_images(65, CONFIGS['parseq'], seed=33).to(DEV)
    x8 = g['images'].to(DEV)
    first65 = _run(default_model, x65, 'ar1')
    first8 = _run(default_model, x8, 'ar1')
    for _ in range(2):
        assert torch.equal(_run(default_model, x65, 'ar1'), first65)
        assert torch.equal(_run(default_model, x8, 'ar1'), first8)
    err, msg = report('small-batch route after a one-launch forward on the same plan', first8, g['logits.ar1'])
    assert err <= 1e-3 and torch.equal(first8.argmax(-1), g['logits.ar1'].argmax(-1)), msg
    # and through encode() + forward on the other route: the K / V cache of an encode() must not survive a forward of another batch
    mem8 = default_model.model.encode(x8)
    assert torch.equal(_run(default_model, x65, 'nar0'), _run(default_model, x65, 'nar0'))
    err, msg = report('memory of the small-batch route after the alternation', mem8.cpu(), g['memory'])
    assert err <= 5e-4, msg
