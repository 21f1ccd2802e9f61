"""Row N1: post-processing.

CPU: the oracle restatement of `tokenizer.decode(logits.softmax(-1))` against the strings / confidences the REFERENCE
tokenizer produced for the golden logits (tests/golden/*.json), and against this package's host-side `Tokenizer.decode`.
GPU: `parseq_postprocess` (through the C ABI) against the oracle — ids and lengths bit-exact, probabilities 1e-6.
"""
import pytest
import torch

from oracle import parseq_oracle as O
from parseq_amd.tokenizer import Tokenizer

CHARSET = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"
MODES = ['nar0', 'nar1', 'ar0', 'ar0_full', 'ar0_len7', 'ar1', 'ar2']


def _strings(tok, ids, lengths):
    return [tok._ids2tok(row[:k].tolist()) for row, k in zip(ids, lengths.tolist())]


def _edge_logits():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(12, 26, 95, generator=g) * 3
    x[0, :, 0] = -50.0                    # no EOS anywhere: length == L, confidence over all L
    x[1, 0, 0] = 50.0                     # EOS first: empty label, confidence = p(EOS)
    x[2, 25, 0] = 50.0; x[2, :25, 0] = -50.0   # EOS at the last position
    x[3, 5, :] = 0.0                      # exact ties: first maximum wins (id 0 = EOS at position 5)
    x[4, 3, 7] = x[4, 3, 9] = 40.0        # two-way tie between characters
    x[5] = x[5] * 30                      # saturated soft-max
    return x


@pytest.mark.parametrize('name', ['parseq', 'parseq-tiny'])
@pytest.mark.parametrize('mode', MODES)
def test_oracle_postprocess_matches_reference_tokenizer(golden, name, mode):
    g, meta = golden(name)
    tok = Tokenizer(CHARSET)
    ids, lengths, probs, conf = O.postprocess(g[f'logits.{mode}'], tok.eos_id)
    assert _strings(tok, ids, lengths) == meta['modes'][mode]['strings']
    want = torch.tensor(meta['modes'][mode]['confidence'])
    assert torch.allclose(conf, want, rtol=1e-5, atol=1e-7)


def test_host_decode_matches_oracle_postprocess():
    tok = Tokenizer(CHARSET)
    x = _edge_logits()
    ids, lengths, probs, conf = O.postprocess(x, tok.eos_id)
    labels, plist = tok.decode(x.softmax(-1))
    assert labels == _strings(tok, ids, lengths)
    for b, p in enumerate(plist):
        n = min(int(lengths[b]) + 1, x.shape[1])
        assert p.shape == (n,) and torch.equal(p, probs[b, :n])
    assert lengths[0] == 26 and lengths[1] == 0 and lengths[2] == 25 and lengths[3] == 5


def test_decode_logits_refuses_cpu():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Tokenizer(CHARSET).decode_logits(torch.zeros(1, 26, 95))


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(12, 26, 95), (1, 1, 95), (5, 64, 95), (3, 8, 37), (513, 26, 95)])
def test_postprocess_kernel_matches_oracle(shape):
    tok = Tokenizer(CHARSET)
    if shape == (12, 26, 95):
        x = _edge_logits()
    else:
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(shape[0] + shape[1])) * 4
        x[..., 0] += 1.5                  # EOS often enough that truncation is exercised
    ids, lengths, probs, conf = O.postprocess(x, tok.eos_id)
    gi, gl, gp, gc = tok._postprocess(x.cuda())
    torch.cuda.synchronize()
    assert torch.equal(gi.cpu().long(), ids)
    assert torch.equal(gl.cpu().long(), lengths)
    assert torch.allclose(gp.cpu(), probs, rtol=2e-6, atol=1e-7)
    assert torch.allclose(gc.cpu(), conf, rtol=2e-5, atol=1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', MODES)
def test_decode_logits_matches_reference_strings(golden, mode):
    g, meta = golden('parseq')
    tok = Tokenizer(CHARSET)
    logits = g[f'logits.{mode}'].cuda()
    labels, plist = tok.decode_logits(logits)
    assert labels == meta['modes'][mode]['strings']
    host_labels, host_plist = tok.decode(logits.softmax(-1))
    assert labels == host_labels
    for a, b in zip(plist, host_plist):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    labels2, conf = tok.read(logits)
    assert labels2 == labels
    assert torch.allclose(conf, torch.tensor(meta['modes'][mode]['confidence']), rtol=2e-5, atol=1e-7)
