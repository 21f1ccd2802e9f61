"""CPU tests of the host-side mirror of the reference interface (hubconf entrypoints, factories, tokenizer, hparams)."""
import pytest
import torch

from oracle.synth import CONFIGS, state_dict_spec


def test_hub_entrypoints_and_signatures():
    import inspect
    m = torch.hub.load('.', 'parseq_tiny', source='local', pretrained=False, decode_ar=False, refine_iters=0)
    assert type(m).__name__ == 'PARSeq' and m.model.decode_ar is False and m.model.refine_iters == 0
    assert sum(p.numel() for p in m.model.parameters()) == 6_018_143
    import hubconf
    for name in ('parseq_tiny', 'parseq', 'parseq_patch16_224'):
        sig = inspect.signature(getattr(hubconf, name))
        assert list(sig.parameters)[:3] == ['pretrained', 'decode_ar', 'refine_iters']
        assert (sig.parameters['pretrained'].default, sig.parameters['decode_ar'].default, sig.parameters['refine_iters'].default) == (False, True, 1)
    assert hubconf.dependencies == ['torch']


@pytest.mark.parametrize('name', ['parseq', 'parseq-tiny', 'parseq-patch16-224'])
def test_state_dict_layout_matches_reference(name):
    from parseq_amd import create_model
    m = create_model(name)
    spec = state_dict_spec(CONFIGS[name])
    sd = m.model.state_dict()
    assert set(sd) == set(spec)
    for k, shape in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    # system-level keys are the inner keys under 'model.' (what a Lightning checkpoint stores)
    assert set(m.state_dict()) == {'model.' + k for k in spec}


def test_hparams_and_attributes():
    from parseq_amd import create_model
    m = create_model('parseq', refine_iters=2, decode_ar=False)
    hp = m.hparams
    assert hp.img_size == [32, 128] and hp.max_label_length == 25 and len(hp.charset_train) == 94
    assert hp.charset_test == '0123456789abcdefghijklmnopqrstuvwxyz'
    assert hp['refine_iters'] == 2 and m.model.refine_iters == 2 and m.model.decode_ar is False
    assert (m.eos_id, m.bos_id, m.pad_id) == (0, 95, 96) and len(m.tokenizer) == 97
    assert m.device.type == 'cpu' and m.eval() is m


def test_factory_errors_and_arg_parsing():
    from parseq_amd import InvalidModelError, create_model, load_from_checkpoint, parse_model_args
    with pytest.raises(InvalidModelError):
        create_model('no-such-model')
    with pytest.raises(InvalidModelError):
        load_from_checkpoint('crnn.ckpt')
    assert parse_model_args(['refine_iters:int=2', 'decode_ar:bool=false', 'lr:float=1e-3', 'name:str=x']) == \
        {'refine_iters': 2, 'decode_ar': False, 'lr': 1e-3, 'name': 'x'}


def test_lightning_style_checkpoint_roundtrip(tmp_path):
    from parseq_amd import create_model, load_from_checkpoint
    m = create_model('parseq-tiny')
    path = tmp_path / 'parseq-tiny.ckpt'
    torch.save({'state_dict': m.state_dict(), 'hyper_parameters': dict(m.hparams)}, path)
    m2 = load_from_checkpoint(str(path), refine_iters=3)
    assert m2.model.refine_iters == 3
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_tokenizer_ids_and_decode_rule():
    from parseq_amd.tokenizer import CharsetAdapter, Tokenizer
    from parseq_amd.configs import CHARSET_94_FULL
    tok = Tokenizer(CHARSET_94_FULL)
    assert (tok.eos_id, tok.bos_id, tok.pad_id) == (0, 95, 96)
    enc = tok.encode(['ab', 'a'])
    assert enc.tolist() == [[95, 11, 12, 0], [95, 11, 0, 96]]
    # greedy decode, cut at the first EOS, probability list keeps the EOS probability
    probs = torch.full((1, 5, 95), 0.001)
    for pos, cls in enumerate([11, 12, 0, 13, 14]):
        probs[0, pos, cls] = 0.9
    labels, ps = tok.decode(probs)
    assert labels == ['ab'] and ps[0].shape == (3,)
    labels, _ = tok.decode(probs[:, :2])
    assert labels == ['ab']
    assert CharsetAdapter('0123456789abcdefghijklmnopqrstuvwxyz')('Ab-C!') == 'abc'


def test_cpu_forward_is_refused():
    from parseq_amd import create_model
    m = create_model('parseq-tiny').eval()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.rand(1, 3, 32, 128))


def test_edit_distance():
    from parseq_amd.system import edit_distance
    assert edit_distance('kitten', 'sitting') == 3 and edit_distance('', 'abc') == 3 and edit_distance('abc', 'abc') == 0


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
    Nothing under parseq_amd/, hubconf.py or read.py may (a product path through the oracle would void every parity claim)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    product = [os.path.join(root, 'hubconf.py'), os.path.join(root, 'read.py')]
    for d, _, files in os.walk(os.path.join(root, 'parseq_amd')):
        product += [os.path.join(d, f) for f in files if f.endswith('.py')]
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    offenders = [p for p in product if pat.search(open(p).read())]
    assert not offenders, offenders
    # bench.py: the only import sits inside cpu_baseline(); __graft_entry__.py: inside smoke()
    for name, func in (('bench.py', 'cpu_baseline'), ('__graft_entry__.py', 'smoke')):
        src = open(os.path.join(root, name)).read()
        for m in pat.finditer(src):
            head = src[:m.start()]
            last_def = re.findall(r'^def (\w+)\(', head, re.M)[-1]
            assert last_def == func, (name, last_def)


@pytest.mark.parametrize('experiment', ['parseq-tiny', 'vitstr'])
def test_pretrained_weights_go_where_the_reference_puts_them(experiment, monkeypatch):
    """`create_model(experiment, pretrained=True)` (strhub/models/utils.py:80-82): the released PARSeq files carry the INNER model's
    keys and are loaded into `system.model`; every other released file (ViTSTR) carries the system's 'model.'-prefixed keys and is
    loaded into the system.  The download is replaced by a state dict of the released files' key layout."""
    import torch
    from parseq_amd import utils
    donor = utils.create_model(experiment)
    gen = torch.Generator().manual_seed(5)
    inner = {k: torch.randn(v.shape, generator=gen) for k, v in donor.model.state_dict().items()}
    released = inner if 'parseq' in experiment else {'model.' + k: v for k, v in inner.items()}
    monkeypatch.setattr(utils, 'get_pretrained_weights', lambda name: released)
    m = utils.create_model(experiment, pretrained=True)
    for k, v in m.model.state_dict().items():
        assert torch.equal(v, inner[k]), k
    wrong = {'model.' + k: v for k, v in inner.items()} if 'parseq' in experiment else inner
    monkeypatch.setattr(utils, 'get_pretrained_weights', lambda name: wrong)
    with pytest.raises(RuntimeError):          # the other layout is refused, as the reference's strict load would
        utils.create_model(experiment, pretrained=True)
