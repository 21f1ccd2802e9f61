"""Models built through the hub keyword arguments the reference accepts (`strhub/models/utils.py:41` `config.update(kwargs)`;
`hubconf.py:13-33` forwards `**kwargs`): other training charsets (`configs/charset/36_lowercase.yaml:3`, `62_mixed-case.yaml:3` —
head 37 / 63 wide instead of 95, embedding 39 / 65 rows instead of 97) and `max_label_length` = 10 (`pos_queries` with 11 rows instead
of 26; every AR / refinement loop, table and mask sized by it).

Goldens: `oracle/make_golden_hub.py` runs the reference's UNMODIFIED `create_model(experiment, **kwargs)` (its YAML resolution, its
system class, its Tokenizer) and stores the resolved configuration, inputs, `memory`, the logits of seven decode modes and the strings
of the system's own tokenizer (tests/golden/<variant>.safetensors / .json).

CPU tests: the configuration resolved here equals the reference's, key by key; the tokenizer's ids; the model built on the CPU has the
reference's parameter shapes; the CPU oracle reproduces the goldens.  GPU tests (through the C ABI): every decode mode in the two
arithmetic modes that meet 1e-3, on each form of the AR step (split over workgroups / one workgroup per row tile / per-op kernels), at
batch 4 and at a batch that fills the device.
"""
import pytest
import torch

from oracle import parseq_oracle as O
from oracle.synth import HUB_VARIANTS, variant_config, variant_state_dict

VARIANTS = list(HUB_VARIANTS)
MODE_NAMES = ['nar0', 'nar1', 'ar0', 'ar0_full', 'ar0_short', 'ar1', 'ar2']


def _build(variant, precision=None, **extra):
    from parseq_amd import create_model
    experiment, kwargs, _ = HUB_VARIANTS[variant]
    if precision is not None:
        extra['precision'] = precision
    m = create_model(experiment, **kwargs, **extra)
    m.model.load_state_dict(variant_state_dict(variant), strict=True)
    return m.eval()


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU
# ---------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('variant', VARIANTS)
def test_resolved_config_equals_the_references(variant, golden):
    """`get_config(experiment, **kwargs)` against `strhub.models.utils._get_config` (utils.py:25-44) run on the reference's YAML."""
    from parseq_amd.configs import get_config
    _, meta = golden(variant)
    experiment, kwargs, _ = HUB_VARIANTS[variant]
    assert meta['experiment'] == experiment and meta['kwargs'] == kwargs
    ours = get_config(experiment, **kwargs)
    ref = meta['resolved_config']
    assert set(ours) == set(ref), set(ours) ^ set(ref)
    for k in ref:
        assert ours[k] == ref[k], (k, ours[k], ref[k])


@pytest.mark.parametrize('variant', VARIANTS)
def test_model_shapes_and_tokenizer_follow_the_kwargs(variant, golden):
    _, meta = golden(variant)
    cfg = variant_config(variant)
    m = _build(variant)
    tok = m.tokenizer
    assert {'len': len(tok), 'eos_id': tok.eos_id, 'bos_id': tok.bos_id, 'pad_id': tok.pad_id} == meta['tokenizer']
    assert (m.bos_id, m.eos_id, m.pad_id) == (cfg.bos_id, cfg.eos_id, cfg.pad_id)
    sd = m.model.state_dict()
    n_cls, n_pos = cfg.num_tokens - 2, cfg.max_label_length + 1
    assert tuple(sd['head.weight'].shape) == (n_cls, cfg.embed_dim) and tuple(sd['head.bias'].shape) == (n_cls,)
    assert tuple(sd['text_embed.embedding.weight'].shape) == (cfg.num_tokens, cfg.embed_dim)
    assert tuple(sd['pos_queries'].shape) == (1, n_pos, cfg.embed_dim)
    assert sum(p.numel() for p in m.model.parameters()) == meta['num_params']
    assert m.hparams['max_label_length'] == cfg.max_label_length and m.hparams['charset_train'] == HUB_VARIANTS[variant][1]['charset_train']
    # test-time charset (configs/main.yaml:11) stays the 36 lower-case characters: the adapter lower-cases when the test set has no upper case
    assert m.charset_adapter('aBc!') == 'abc'


@pytest.mark.parametrize('variant', VARIANTS)
def test_oracle_reproduces_the_hub_kwarg_goldens(variant, golden):
    g, meta = golden(variant)
    cfg, sd = variant_config(variant), variant_state_dict(variant)
    with torch.inference_mode():
        mem = O.encode(sd, cfg, g['images'])
        assert (mem - g['memory']).abs().max().item() <= 5e-6
        for mode in MODE_NAMES:
            spec = meta['modes'][mode]
            got = O.forward(sd, cfg, g['images'], spec['max_length'], decode_ar=spec['decode_ar'], refine_iters=spec['refine_iters'])
            ref = g[f'logits.{mode}']
            assert list(got.shape) == spec['shape']
            assert (got - ref).abs().max().item() <= 5e-6, (variant, mode)


# ---------------------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------------------

DEV = 'cuda'


def _run(m, images, spec, **kw):
    m.model.decode_ar, m.model.refine_iters = spec['decode_ar'], spec['refine_iters']
    with torch.inference_mode():
        out = m(images, spec['max_length'], **kw)
    torch.cuda.synchronize()
    return out.float().cpu()


def _check(tag, m, got, g, meta, mode, tol=1e-3):
    from gpu_util import report
    ref = g[f'logits.{mode}']
    assert list(got.shape) == list(ref.shape) == meta['modes'][mode]['shape'], (tag, got.shape, ref.shape)
    err, msg = report(f'{tag} {mode} logits vs reference', got, ref)
    assert err <= tol, msg
    assert torch.equal(got.argmax(-1), ref.argmax(-1)), msg
    strings, _ = m.tokenizer.decode(got.softmax(-1))
    assert strings == meta['modes'][mode]['strings'], msg


@pytest.fixture(scope='module')
def variant_models():
    cache = {}

    def get(variant, precision):
        if (variant, precision) not in cache:
            cache[(variant, precision)] = _build(variant, precision).to(DEV)
        return cache[(variant, precision)]
    return get


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('mode', MODE_NAMES)
@pytest.mark.parametrize('variant', VARIANTS)
def test_forward_matches_reference(variant, mode, precision, variant_models, golden):
    """The seven decode modes: <= 1e-3 on the logits, arg-max identical, strings identical — the default call (a forward that has the
    device to itself: the AR step split over workgroups)."""
    g, meta = golden(variant)
    m = variant_models(variant, precision)
    got = _run(m, g['images'].to(DEV), meta['modes'][mode])
    _check(f'{variant} {precision}', m, got, g, meta, mode)


@pytest.mark.gpu
@pytest.mark.parametrize('variant', VARIANTS)
def test_encoder_memory(variant, variant_models, golden):
    from gpu_util import report
    g, _ = golden(variant)
    for precision, tol in (('fp32', 2e-4), ('bf16x3', 5e-4)):
        mem = variant_models(variant, precision).model.encode(g['images'].to(DEV)).cpu()
        err, msg = report(f'{variant} memory {precision}', mem, g['memory'])
        assert err <= tol, msg


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
@pytest.mark.parametrize('variant', VARIANTS)
def test_every_form_of_the_ar_step(variant, precision, variant_models, golden, monkeypatch):
    """The AR loop's three forms at C = 37 / 63 classes and 11 positions: split over workgroups (default call), one workgroup per row
    tile (an explicit slot: batches in flight), the per-op kernels (PARSEQ_NO_FUSED_STEP=1, read when a plan is created)."""
    from gpu_util import report
    g, meta = golden(variant)
    images = g['images'].to(DEV)
    m = variant_models(variant, precision)
    monkeypatch.setenv('PARSEQ_NO_FUSED_STEP', '1')
    perop = _build(variant, precision).to(DEV)
    monkeypatch.delenv('PARSEQ_NO_FUSED_STEP')
    for mode in ('ar0', 'ar0_full', 'ar1', 'ar2'):
        spec = meta['modes'][mode]
        outs = {'split': _run(m, images, spec), 'one-workgroup': _run(m, images, spec, slot=0), 'per-op': _run(perop, images, spec)}
        for form, got in outs.items():
            if precision == 'bf16x3':
                _check(f'{variant} bf16x3 AR step [{form}]', m, got, g, meta, mode)
            else:
                # bf16 operands: 6e-2 of the exact reference (tests/test_hip_parity.py's bar), decisions identical up to the first near-tie
                ref = g[f'logits.{mode}']
                assert list(got.shape) == list(ref.shape)
                err, msg = report(f'{variant} bf16 AR step [{form}] {mode}', got, ref)
                top2 = ref.topk(2, -1).values
                safe = ((top2[..., 0] - top2[..., 1]) > 0.12).int().cumprod(-1).bool()
                assert torch.isfinite(got).all() and bool((got.argmax(-1) == ref.argmax(-1))[safe].all()), msg
                assert (got - ref).abs()[safe].max().item() <= 6e-2 if bool(safe.any()) else True, msg


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('variant', VARIANTS)
def test_device_filling_and_ragged_batches(variant, precision, variant_models, golden):
    """The goldens' crops tiled to batches that fill the device and to ragged row tiles: every copy must equal the batch-of-4 result
    (same image alone or in a batch) — tables, head tiles and the EOS counter at other strides than 26 x 97."""
    g, meta = golden(variant)
    m = variant_models(variant, precision)
    images = g['images'].to(DEV)
    tol = {'fp32': 1e-5, 'bf16x3': 1e-4, 'bf16': 0.0}[precision]      # bf16x3: the GEMM tiling and the AR step's split follow the batch size (summation order)
    for mode in ('ar1', 'ar0_full', 'nar1'):
        spec = meta['modes'][mode]
        small = _run(m, images, spec)
        for B in (512, 37):
            big = _run(m, images.repeat((B + 3) // 4, 1, 1, 1)[:B].contiguous(), spec)
            want = small.repeat((B + 3) // 4, 1, 1)[:B]
            d = (big - want).abs().max().item()
            assert big.shape == want.shape and d <= tol, (variant, precision, mode, B, d)
        if precision != 'bf16':
            _check(f'{variant} {precision} first rows of batch 37', m, big[:4], g, meta, mode)


@pytest.mark.gpu
@pytest.mark.parametrize('variant', VARIANTS)
def test_postprocess_and_test_step_on_the_variant(variant, variant_models, golden):
    """Row N1 (device post-process) and `test_step` (base.py:179-180) with the narrower head: strings and confidences of the reference's
    tokenizer on the reference's logits."""
    g, meta = golden(variant)
    m = variant_models(variant, 'bf16x3')
    spec = meta['modes']['ar1']
    m.model.decode_ar, m.model.refine_iters = True, 1
    with torch.inference_mode():
        logits = m(g['images'].to(DEV))
        strings, conf = m.tokenizer.read(logits)
    assert list(strings) == spec['strings']
    assert torch.allclose(conf.float().cpu(), torch.tensor(spec['confidence']), rtol=2e-3, atol=1e-6)
    labels = [m.charset_adapter(s) for s in spec['strings']]
    out = m.test_step((g['images'].to(DEV), labels), 0)['output']
    assert out.num_samples == 4 and out.correct == 4


# ---------------------------------------------------------------------------------------------------------------------------------
# ViTSTR through the same keyword arguments (strhub/models/vitstr/system.py:41-60)
# ---------------------------------------------------------------------------------------------------------------------------------

def _build_vitstr(variant, precision=None):
    from oracle import vitstr_oracle as V
    from oracle.synth import VITSTR_HUB_VARIANTS, vitstr_variant_config
    from parseq_amd import create_model
    extra = {'precision': precision} if precision else {}
    m = create_model('vitstr', **VITSTR_HUB_VARIANTS[variant], **extra)
    m.model.load_state_dict(V.synth_state_dict(vitstr_variant_config(variant), 0), strict=True)
    return m.eval()


def test_vitstr_variant_config_shapes_and_oracle(golden):
    from oracle import vitstr_oracle as V
    from oracle.synth import VITSTR_HUB_VARIANTS, vitstr_variant_config
    from parseq_amd.configs import get_config
    for variant, kwargs in VITSTR_HUB_VARIANTS.items():
        g, meta = golden(variant)
        ours, ref = get_config('vitstr', **kwargs), meta['resolved_config']
        assert set(ours) == set(ref) and all(ours[k] == ref[k] for k in ref)
        cfg = vitstr_variant_config(variant)
        m = _build_vitstr(variant)
        assert {'len': len(m.tokenizer), 'eos_id': m.tokenizer.eos_id, 'bos_id': m.tokenizer.bos_id, 'pad_id': m.tokenizer.pad_id} == meta['tokenizer']
        assert tuple(m.model.state_dict()['head.weight'].shape) == (cfg.num_tokens - 2, cfg.embed_dim)
        assert sum(p.numel() for p in m.model.parameters()) == meta['num_params']
        sd = V.synth_state_dict(cfg, 0)
        with torch.inference_mode():
            assert (V.forward(sd, cfg, g['images']) - g['logits']).abs().max().item() <= 5e-6
            assert (V.forward(sd, cfg, g['images'], meta['short_max_length']) - g['logits.short']).abs().max().item() <= 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16'])
def test_vitstr_variant_matches_reference(precision, golden):
    from gpu_util import report
    from oracle.synth import VITSTR_HUB_VARIANTS
    for variant in VITSTR_HUB_VARIANTS:
        g, meta = golden(variant)
        m = _build_vitstr(variant, precision).to(DEV)
        x = g['images'].to(DEV)
        with torch.inference_mode():
            full, short = m(x).float().cpu(), m(x, meta['short_max_length']).float().cpu()
            big = m(x.repeat(16, 1, 1, 1)).float().cpu()
        assert list(full.shape) == meta['shapes']['logits'] and list(short.shape) == meta['shapes']['logits.short']
        tol = 1e-3 if precision != 'bf16' else 6e-2
        for tag, got, ref in (('full', full, g['logits']), ('short', short, g['logits.short']), ('batch 64, first rows', big[:4], g['logits'])):
            err, msg = report(f'{variant} {precision} {tag}', got, ref)
            assert err <= tol, msg
            if precision != 'bf16':
                assert torch.equal(got.argmax(-1), ref.argmax(-1)), msg
        if precision != 'bf16':
            strings, _ = m.tokenizer.decode(full.softmax(-1))
            assert strings == meta['strings']
