"""Row N3 (training step), forward half.

CPU: the permutation sampler and mask construction of parseq_amd/system.py (host logic) and the oracle's `training_loss`
against golden vectors minted by executing the reference's own `strhub/models/parseq/system.py` (oracle/make_golden_train.py),
plus the oracle's autograd gradients against the reference's — the gate the backward kernels will be held to.
GPU: the device evaluation of the K-permutation loss (`parseq_decode_logits` + `parseq_cross_entropy` per permutation)
against the oracle on the same crops, labels and permutations."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import parseq_oracle as O
from oracle.synth import CONFIGS, synth_images, synth_state_dict
from parseq_amd.system import gen_tgt_perms, generate_attn_masks
from parseq_amd.tokenizer import Tokenizer

CHARSET_94 = ("0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
              "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")


@pytest.fixture(scope='module')
def perm_golden():
    with open(os.path.join(GOLDEN_DIR, 'perms.json')) as f:
        return json.load(f)


def _sample(case):
    rng = np.random.default_rng(case['np_seed'])
    torch.manual_seed(case['torch_seed'])
    max_gen = case['perm_num'] // 2 if case['perm_mirrored'] else case['perm_num']
    return gen_tgt_perms(torch.zeros(3, case['num_chars'] + 2, dtype=torch.long), max_gen, case['perm_forward'],
                         case['perm_mirrored'], rng)


def test_permutation_sampler_matches_reference(perm_golden):
    """Every label length 1..25 x six (perm_num, perm_forward, perm_mirrored) settings, same numpy / torch seeds: identical
    orderings (pool branch below five characters, torch.randperm above, the one-character special case)."""
    if perm_golden['torch'] != torch.__version__ or perm_golden['numpy'] != np.__version__:
        pytest.skip('seeded streams are only comparable under the torch / numpy builds the goldens were minted with')
    assert len(perm_golden['perms']) == 147
    for case in perm_golden['perms']:
        assert _sample(case).tolist() == case['perms'], case


@pytest.mark.parametrize('setting', [(6, True, True), (6, False, True), (5, True, False), (1, True, False), (24, True, True)])
def test_permutation_sampler_properties(setting):
    """Seed-independent facts of system.py:90-150: every row is an ordering of 0..T+1 that starts at <bos>; the count is
    min(perm_num, T! [/ 2 * 2]); mirrored pairs are reverses of each other; row 1 is right-to-left with <eos> first; below
    five characters no ordering repeats."""
    perm_num, fwd, mir = setting
    rng = np.random.default_rng(5)
    for T in range(1, 26):
        p = gen_tgt_perms(torch.zeros(2, T + 2), perm_num // 2 if mir else perm_num, fwd, mir, rng)
        if T == 1:
            assert p.tolist() == [[0, 1, 2]]
            continue
        full = math.factorial(T)
        assert len(p) == (2 * min(perm_num // 2, full // 2) if mir else min(perm_num, full))
        assert (p.sort(dim=1).values == torch.arange(T + 2)).all() and (p[:, 0] == 0).all()
        if fwd:
            assert p[0].tolist() == list(range(T + 2))
        if len(p) > 1:
            assert p[1].tolist() == [0] + list(range(T + 1, 0, -1))
        rest = torch.cat([p[:1], p[2:]])
        assert (rest[:, -1] == T + 1).all()                        # everywhere else <eos> is generated last
        if mir:
            chars = p[:, 1:-1]
            assert torch.equal(chars[2::2].flip(-1), chars[3::2])  # (pair 0 is forward / the rewritten row 1)
        if T < 5:
            assert len({tuple(r) for r in p.tolist()}) == len(p)


def test_attention_masks_match_reference(perm_golden):
    for case in perm_golden['masks']:
        perm = torch.tensor(case['perm'])
        for fn in (generate_attn_masks, O.attn_masks_from_perm):     # host logic and the oracle's restatement
            cm, qm = fn(perm)
            assert cm.dtype == qm.dtype == torch.bool
            assert cm.int().tolist() == case['content_mask'] and qm.int().tolist() == case['query_mask']
    # forward ordering: the query mask is the strict-causal mask shifted by one (query i sees <bos> .. token i-1 ... i.e. keys < i+1)
    _, qm = generate_attn_masks(torch.arange(8))
    assert torch.equal(qm, torch.triu(torch.ones(7, 7, dtype=torch.bool), diagonal=1))


@pytest.fixture(scope='module')
def train_golden(golden):
    return golden('parseq_train')


def test_oracle_training_loss_matches_reference(train_golden):
    """One training step of the reference (dropout off) on the synthetic PARSeq-S: the oracle's loss equals the reference's
    to fp32 round-off, the per-permutation target counts follow the <eos>-dropping rule, and autograd through the oracle
    reproduces the reference's gradient for every one of the 175 parameters."""
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    sd = {k: v.requires_grad_(True) for k, v in synth_state_dict(cfg, 0).items()}
    tok = Tokenizer(CHARSET_94)
    tgt = tok.encode(meta['labels'])
    assert torch.equal(g['images'], synth_images(len(meta['labels']), cfg, seed=meta['image_seed']))
    perms = g['perms'].long()
    assert perms.tolist() == meta['perms'] and perms.shape == (6, 27)
    loss, per_perm, counts = O.training_loss(sd, cfg, g['images'], tgt, perms)
    assert abs(float(loss.detach()) - meta['loss']) <= 2e-6 * meta['loss']
    chars = sum(len(s) for s in meta['labels'])
    assert counts.tolist() == [chars + 8] * 2 + [chars] * 4
    loss.backward()
    assert len(meta['grads']) == 175
    for k, want in meta['grads'].items():
        grad = sd[k].grad
        assert (grad is None) == want['none'], k
        got = float(grad.double().norm()) if grad is not None else 0.0
        assert abs(got - want['norm']) <= 1e-4 * max(want['norm'], 1e-4), (k, got, want['norm'])
    for k, want in g.items():
        if k.startswith('grad.'):
            torch.testing.assert_close(sd[k[5:]].grad, want, rtol=0, atol=1e-6 * max(1.0, float(want.abs().max()) * 100))


def test_training_loss_reduction_rule():
    """system.py:184-196 on hand-made numbers: the loss is the target-count-weighted mean of the per-permutation means, with
    the counts of permutations 2.. excluding <eos>."""
    cfg = CONFIGS['parseq-tiny']
    sd = synth_state_dict(cfg, 3)
    tok = Tokenizer(CHARSET_94)
    tgt = tok.encode(['abc', 'defgh', 'i'])
    perms = gen_tgt_perms(tgt, 3, True, True, np.random.default_rng(0))
    with torch.inference_mode():
        loss, per_perm, counts = O.training_loss(sd, cfg, synth_images(3, cfg, 9), tgt, perms)
    assert counts.tolist() == [12, 12, 9, 9, 9, 9]
    want = float((per_perm.double() * counts).sum() / counts.sum())
    assert abs(float(loss) - want) <= 1e-6 * want


def test_hand_derived_decoder_backward_matches_reference(train_golden):
    """oracle/decoder_backward.py — the chain rule spelled out in the order the device launches its kernels — against the
    reference's `loss.backward()` for every decoder-side parameter, and against autograd for the gradient w.r.t. memory."""
    from oracle import decoder_backward as DB
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    tgt = Tokenizer(CHARSET_94).encode(meta['labels'])
    perms = g['perms'].long()
    with torch.no_grad():
        memory = O.encode(sd, cfg, g['images'])
        loss, per_perm, grads, dmem = DB.loss_and_grads(sd, cfg, memory, tgt, perms, O.attn_masks_from_perm)
    assert abs(float(loss) - meta['loss']) <= 2e-6 * meta['loss']
    assert set(grads) == {k for k in meta['grads'] if not k.startswith('encoder.')} and len(grads) == 26
    for k, got in grads.items():
        want = meta['grads'][k]
        assert abs(float(got.double().norm()) - want['norm']) <= 1e-4 * max(want['norm'], 1e-6), k
        if 'grad.' + k in g:
            assert (got - g['grad.' + k]).abs().max() <= 1e-5 * float(g['grad.' + k].abs().max()) + 1e-8, k
    # d loss / d memory: autograd through the oracle's decoder with memory as the leaf
    mem = memory.clone().requires_grad_(True)
    sd_mem = dict(sd)
    tgt_in, tgt_out = tgt[:, :-1], tgt[:, 1:]
    pad = (tgt_in == cfg.pad_id) | (tgt_in == cfg.eos_id)
    total, numel, n = 0.0, 0, int((tgt_out != cfg.pad_id).sum())
    for i, perm in enumerate(perms):
        out = O.decode(sd_mem, cfg, tgt_in, mem, None, pad, tgt_query_mask=O.attn_masks_from_perm(perm)[1])
        ce = torch.nn.functional.cross_entropy(O.head(sd_mem, out).flatten(end_dim=1), tgt_out.flatten(), ignore_index=cfg.pad_id)
        total, numel = total + n * ce, numel + n
        if i == 1:
            tgt_out = torch.where(tgt_out == cfg.eos_id, cfg.pad_id, tgt_out)
            n = int((tgt_out != cfg.pad_id).sum())
    (total / numel).backward()
    assert (mem.grad - dmem).abs().max() <= 1e-5 * float(mem.grad.abs().max())


@pytest.mark.parametrize('name', ['parseq-tiny', 'parseq'])
def test_hand_derived_encoder_backward_matches_autograd(name):
    """oracle/encoder_backward.py (forward that keeps x / qkv / attention output / x_mid / fc1 pre-activation per block, backward
    from d memory) against autograd through the oracle's own `encode`, for an arbitrary upstream gradient."""
    from oracle import encoder_backward as EB
    cfg = CONFIGS[name]
    sd = synth_state_dict(cfg, 2)
    images = synth_images(3, cfg, seed=5)
    upstream = torch.randn(3, cfg.num_patches, cfg.embed_dim, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        memory, saved = EB.forward(sd, cfg, images)
        grads = EB.backward(sd, cfg, saved, upstream)
    leaves = {k: (v.clone().requires_grad_(True) if k.startswith('encoder.') else v) for k, v in sd.items()}
    mem = O.encode(leaves, cfg, images)
    assert (mem.detach() - memory).abs().max() <= 2e-5
    (mem * upstream).sum().backward()
    assert len(grads) == sum(k.startswith('encoder.') for k in sd)
    for k, got in grads.items():
        want = leaves[k].grad
        assert (got - want).abs().max() <= 5e-5 * float(want.abs().max()) + 1e-7, k


def test_loss_denominator_matches_the_counted_targets():
    """parseq_amd.train.loss_denominator (computed from the labels on the host, so the step needs no device read) against the
    counts `training_step` derives from the encoded targets (system.py:183-196), for 1..8 permutation passes."""
    from parseq_amd.train import loss_denominator
    cfg = CONFIGS['parseq-tiny']
    sd = synth_state_dict(cfg, 3)
    tok = Tokenizer(CHARSET_94)
    labels = ['a', 'bc', 'hello', 'W0rld#42', 'x' * 25, 'Zz']
    tgt = tok.encode(labels)
    for k in range(1, 9):
        perms = torch.stack([torch.arange(tgt.shape[1])] * k)
        with torch.inference_mode():
            counts = O.training_loss(sd, cfg, synth_images(len(labels), cfg, 1), tgt, perms)[2]
        assert int(counts.sum()) == loss_denominator(labels, k), k


def test_bf16_operand_rounding_budget(train_golden):
    """What rounding every matrix-product operand to bfloat16 (fp32 accumulate, everything else fp32) does to one training step
    of PARSeq-S: the tolerance a bf16 device path (BASELINE configs[4]) can be held to against the fp32 gate.  Measured:
    loss 4e-5 relative; per-tensor gradient error 0.8 % median, 2.8 % worst (L2, relative); cosine >= 0.9996."""
    from oracle import decoder_backward as DB, encoder_backward as EB
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    tgt = Tokenizer(CHARSET_94).encode(meta['labels'])
    perms = g['perms'].long()

    def step():
        with torch.no_grad():
            memory, saved = EB.forward(sd, cfg, g['images'])
            loss, _, grads, dmem = DB.loss_and_grads(sd, cfg, memory, tgt, perms, O.attn_masks_from_perm)
            grads.update(EB.backward(sd, cfg, saved, dmem))
        return float(loss), grads

    exact_loss, exact = step()
    with DB.rounding('bf16'):
        bf_loss, bf = step()
    assert DB._ROUNDING[0] is None
    assert abs(exact_loss - meta['loss']) <= 2e-6 * meta['loss'] and 0 < abs(bf_loss - exact_loss) <= 5e-4 * exact_loss
    rel, cos = [], []
    for k, a in exact.items():
        a, b = a.double().flatten(), bf[k].double().flatten()
        if float(a.norm()) < 1e-7:
            continue
        rel.append(float((a - b).norm() / a.norm()))
        cos.append(float(a @ b / (a.norm() * b.norm())))
    rel.sort()
    assert 1e-3 < rel[len(rel) // 2] < 2e-2 and rel[-1] < 6e-2 and min(cos) > 0.998


def test_bf16_operand_budget_against_bf16_mixed_autocast(train_golden):
    """The reference trains under `precision: bf16-mixed` (configs/main.yaml, train.py:62-64 — torch.autocast), where a Linear / matmul ROUNDS
    ITS OUTPUT to bfloat16 as well as its operands.  The device's bf16-operand mode is gated by a budget against the EXACT fp32 backward
    (test_bf16_operand_rounding_budget); this test holds that budget against what autocast itself does to the same step: autograd through
    the oracle's training_loss under torch.autocast('cpu', bfloat16) — (1) the operand-rounding model is at least as close to the exact
    gradients as autocast is (it keeps fp32 outputs), tensor by tensor in the median and in the worst case; (2) the two reduced-precision
    gradients are as close to each other as their distances to the exact ones allow."""
    from oracle import decoder_backward as DB, encoder_backward as EB
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    tgt = Tokenizer(CHARSET_94).encode(meta['labels'])
    perms = g['perms'].long()

    def hand(rounding):
        ctx = DB.rounding(rounding) if rounding else __import__('contextlib').nullcontext()
        with torch.no_grad(), ctx:
            memory, saved = EB.forward(sd, cfg, g['images'])
            loss, _, grads, dmem = DB.loss_and_grads(sd, cfg, memory, tgt, perms, O.attn_masks_from_perm)
            grads.update(EB.backward(sd, cfg, saved, dmem))
        return float(loss), grads

    exact_loss, exact = hand(None)
    ours_loss, ours = hand('bf16')
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    with torch.autocast('cpu', dtype=torch.bfloat16):
        ac_loss, _, _ = O.training_loss(leaf, cfg, g['images'], tgt, perms)
    ac_loss.float().backward()
    ac = {k: v.grad.float() for k, v in leaf.items() if v.grad is not None}
    assert abs(float(ac_loss.detach()) - exact_loss) <= 2e-2 * exact_loss and abs(ours_loss - exact_loss) <= 5e-4 * exact_loss
    rel = lambda a, b: float((a.double() - b.double()).norm() / a.double().norm())
    ours_e, ac_e, between = [], [], []
    for k, a in exact.items():
        if float(a.double().norm()) < 1e-7 or k not in ac:
            continue
        ours_e.append(rel(a, ours[k])); ac_e.append(rel(a, ac[k])); between.append(rel(ac[k], ours[k]))
    assert len(ours_e) >= 170
    med = lambda v: sorted(v)[len(v) // 2]
    print(f'bf16-operand vs exact: median {med(ours_e):.2e} worst {max(ours_e):.2e}; autocast vs exact: median {med(ac_e):.2e} worst {max(ac_e):.2e}; '
          f'between them: median {med(between):.2e} worst {max(between):.2e}')
    assert med(ours_e) <= med(ac_e) and max(ours_e) <= max(ac_e)                 # (1): never looser than the reference's own precision mode
    assert med(ours_e) < 2e-2 and max(ours_e) < 6e-2                             # the budget the device gate uses
    assert all(b <= 1.5 * (o + a) + 1e-6 for b, o, a in zip(between, ours_e, ac_e))      # (2): triangle inequality with slack for the norm's base


def _autograd_decoder_loss(sd, cfg, memory, tgt, perms, drop):
    """The training loss with dropout masks from `drop`, written with plain differentiable torch ops (F.layer_norm, F.softmax,
    F.gelu, F.embedding) — independent of the operator code in oracle/decoder_backward.py — for autograd to differentiate."""
    import torch.nn.functional as F
    from oracle import decoder_backward as DB
    E, H = cfg.embed_dim, cfg.dec_num_heads
    p = 'decoder.layers.0.'
    B, L = tgt.shape[0], tgt.shape[1] - 1
    S = memory.shape[1]
    tgt_in, tgt_out = tgt[:, :-1], tgt[:, 1:]
    pad = (tgt_in == cfg.pad_id) | (tgt_in == cfg.eos_id)
    norm = lambda x, k: F.layer_norm(x, (E,), sd[k + '.weight'], sd[k + '.bias'], 1e-5)
    heads = lambda x, n: x.view(B, n, H, 32).transpose(1, 2)

    def mha(pre, q_in, kv_in, n_kv, mask, pf):
        w, b = sd[pre + 'in_proj_weight'], sd[pre + 'in_proj_bias']
        q, k, v = heads(F.linear(q_in, w[:E], b[:E]), L), heads(F.linear(kv_in, w[E:2 * E], b[E:2 * E]), n_kv), heads(F.linear(kv_in, w[2 * E:], b[2 * E:]), n_kv)
        s_ = q @ k.transpose(-1, -2) / (32 ** 0.5)
        if mask is not None:
            s_ = s_.masked_fill(mask.unsqueeze(1), float('-inf'))
        o = (F.softmax(s_, -1) * pf) @ v
        return F.linear(o.transpose(1, 2).reshape(B, L, E), sd[pre + 'out_proj.weight'], sd[pre + 'out_proj.bias'])

    pq = sd['pos_queries'][:, :L]
    total, numel, n = 0.0, 0, int((tgt_out != cfg.pad_id).sum())
    for i, perm in enumerate(perms):
        f = lambda s_, shape: drop.factor(8 * i + s_, shape)
        emb = (E ** 0.5) * F.embedding(tgt_in, sd['text_embed.embedding.weight'])
        content = torch.cat([emb[:, :1], emb[:, 1:] + pq[:, :L - 1]], dim=1) * f(DB.S_CONTENT, (B, L, E))
        x = pq.expand(B, -1, -1) * f(DB.S_QUERY, (B, L, E))
        mask = O.attn_masks_from_perm(perm)[1].unsqueeze(0) | pad.unsqueeze(1)
        x = x + mha(p + 'self_attn.', norm(x, p + 'norm_q'), norm(content, p + 'norm_c'), L, mask, f(DB.S_SA_PROB, (B, H, L, L))) * f(DB.S_SA_OUT, (B, L, E))
        x = x + mha(p + 'cross_attn.', norm(x, p + 'norm1'), memory, S, None, f(DB.S_CA_PROB, (B, H, L, S))) * f(DB.S_CA_OUT, (B, L, E))
        h = F.gelu(F.linear(norm(x, p + 'norm2'), sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])) * f(DB.S_FF_HIDDEN, (B, L, 4 * E))
        x = x + F.linear(h, sd[p + 'linear2.weight'], sd[p + 'linear2.bias']) * f(DB.S_FF_OUT, (B, L, E))
        logits = F.linear(norm(x, 'decoder.norm'), sd['head.weight'], sd['head.bias'])
        total = total + n * F.cross_entropy(logits.flatten(end_dim=1), tgt_out.flatten(), ignore_index=cfg.pad_id)
        numel += n
        if i == 1:
            tgt_out = torch.where(tgt_out == cfg.eos_id, cfg.pad_id, tgt_out)
            n = int((tgt_out != cfg.pad_id).sum())
    return total / numel


@pytest.mark.parametrize('p_drop', [0.0, 0.1])
def test_hand_derived_backward_with_dropout_matches_autograd(train_golden, p_drop):
    """The eight dropout sites of a permutation pass (model.py:99-102, modules.py:33-43,70-79) with masks from the counter-based
    generator: hand-derived gradients == autograd through an independently written forward using the same masks."""
    from oracle import decoder_backward as DB
    g, meta = train_golden
    cfg = CONFIGS['parseq-tiny']
    sd = {k: v for k, v in synth_state_dict(cfg, 1).items() if not k.startswith('encoder.')}
    tgt = Tokenizer(CHARSET_94).encode(meta['labels'][:5])
    torch.manual_seed(5)
    perms = gen_tgt_perms(tgt, 2, True, True, np.random.default_rng(1))
    assert perms.shape == (4, tgt.shape[1])
    memory = torch.randn(5, 128, cfg.embed_dim, generator=torch.Generator().manual_seed(3))
    drop = DB.Dropout(p_drop, seed=0x1234567890ABCDEF)
    if p_drop:
        f = drop.factor(3, (200, 1000))
        u = f.unique().tolist()
        assert len(u) == 2 and u[0] == 0.0 and abs(u[1] - 1 / 0.9) < 1e-6 and abs(float((f == 0).float().mean()) - 0.1) < 4e-3
        assert not torch.equal(f, drop.factor(4, (200, 1000)))                      # sites are independent streams
    with torch.no_grad():
        loss, _, grads, dmem = DB.loss_and_grads(sd, cfg, memory, tgt, perms, O.attn_masks_from_perm, dropout=drop)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    mem = memory.clone().requires_grad_(True)
    want = _autograd_decoder_loss(leaves, cfg, mem, tgt, perms, drop)
    want.backward()
    assert abs(float(loss) - float(want.detach())) <= 1e-5 * float(want.detach())
    for k, v in leaves.items():
        assert (grads[k] - v.grad).abs().max() <= 2e-5 * float(v.grad.abs().max()) + 1e-8, k
    assert (dmem - mem.grad).abs().max() <= 2e-5 * float(mem.grad.abs().max())


@pytest.mark.parametrize('total,pct', [(100, 0.075), (1000, 0.3), (37, 0.075), (20, 0.5)])
def test_one_cycle_schedule_matches_torch(total, pct):
    """The schedule of base.py:103-106 (OneCycleLR, cycle_momentum=False) restated as a pure function of the step count."""
    from parseq_amd.train import one_cycle_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, 2.5e-3, total, pct_start=pct, cycle_momentum=False)
    for step in range(total):
        assert abs(one_cycle_lr(step, total, 2.5e-3, pct) - opt.param_groups[0]['lr']) <= 1e-9 * 2.5e-3 + 1e-15, step
        opt.step()
        if step + 1 < total:
            sched.step()
    with pytest.raises(ValueError):
        one_cycle_lr(total, total, 2.5e-3, pct)


# ---- device ---------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('bf16', 2e-2)])
def test_permutation_loss_on_device(train_golden, precision, tol):
    """The device evaluation (one encode, six masked teacher-forced decodes, six cross-entropies, no host sync) against the
    reference's loss for the same crops / labels / permutations, and per permutation against the oracle."""
    from gpu_util import DEV, make_model
    from parseq_amd.system import permutation_loss
    g, meta = train_golden
    m = make_model('parseq', precision)
    perms = g['perms'].long()
    loss, per_perm, counts, _ = permutation_loss(m, g['images'].to(DEV), meta['labels'], perms)
    torch.cuda.synchronize()
    assert not loss.requires_grad
    cfg = CONFIGS['parseq']
    with torch.inference_mode():
        _, want_pp, want_counts = O.training_loss(synth_state_dict(cfg, 0), cfg, g['images'], m.tokenizer.encode(meta['labels']), perms)
    assert counts.cpu().tolist() == want_counts.tolist()
    assert (per_perm.cpu() - want_pp).abs().max() <= tol * float(want_pp.max())
    assert abs(float(loss) - meta['loss']) <= tol * meta['loss']


@pytest.mark.gpu
def test_training_step_draws_permutations_and_is_repeatable(train_golden):
    """`training_step` = sampler + loss: with the sampler state rewound the same loss comes back bit for bit; short labels
    (pool branch) and one-character labels (single ordering) go through the same path."""
    from gpu_util import DEV, make_model
    g, meta = train_golden
    m = make_model('parseq', 'bf16')
    images = g['images'].to(DEV)
    out = []
    for _ in range(2):
        m.rng = np.random.default_rng(3)
        torch.manual_seed(4)
        with torch.no_grad():
            out.append(float(m.training_step((images, meta['labels']), 0)))
    assert out[0] == out[1] and math.isfinite(out[0]) and 3.0 < out[0] < 8.0
    # with autograd on, training_step is one node whose backward fills .grad of every parameter (the reference's contract)
    m.rng = np.random.default_rng(3)
    torch.manual_seed(4)
    m.model.zero_grad()
    loss = m.training_step((images, meta['labels']), 0)
    assert loss.requires_grad and abs(float(loss.detach()) - out[0]) <= 2e-2 * out[0]
    (2.0 * loss).backward()
    from parseq_amd.train import loss_and_grads
    m.rng = np.random.default_rng(3)
    torch.manual_seed(4)
    ref = loss_and_grads(m, images, meta['labels'])
    for key, prm in m.model.named_parameters():
        assert prm.grad is not None and torch.equal(prm.grad, 2.0 * ref.grads[key]), key
    for labels in (['ab', 'c', 'abcd', 'xyz', 'q', 'rs', 'tuv', 'w'], ['a', 'b', 'c', 'd', 'e', 'f', 'g', 'h']):
        m.rng = np.random.default_rng(3)
        tgt = m.tokenizer.encode(labels)
        perms = m.gen_tgt_perms(tgt)
        m.rng = np.random.default_rng(3)
        with torch.no_grad():
            got = float(m.training_step((images, labels), 0))
        cfg = CONFIGS['parseq']
        with torch.inference_mode():
            want = float(O.training_loss(synth_state_dict(cfg, 0), cfg, g['images'], tgt, perms)[0])
        assert abs(got - want) <= 2e-2 * want, (labels, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('p_drop', [0.0, 0.1])
def test_decoder_backward_matches_reference_gradients(train_golden, p_drop):
    """`parseq_train_decoder`: loss, gradient of every decoder-side parameter, gradient w.r.t. the encoder output and every
    intermediate of the last permutation against the hand-derived CPU backward (oracle/decoder_backward.py, itself checked
    against autograd).  Dropout off: also against the REFERENCE's `loss.backward()` (tests/golden/parseq_train.*).  Dropout
    0.1: the CPU side regenerates the very same masks (the counter-based generator restated), so the comparison stays exact."""
    from gpu_util import DEV, make_model
    from oracle import decoder_backward as DB
    from parseq_amd.train import decoder_backward
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    m = make_model('parseq', 'fp32')
    perms = g['perms'].long()
    seed = 0x0123456789ABCDEF
    res = decoder_backward(m, g['images'].to(DEV), meta['labels'], perms, dropout=p_drop, seed=seed)
    torch.cuda.synchronize()
    with torch.no_grad():
        memory = O.encode(sd, cfg, g['images'])
        trace = {}
        want_loss, want_pp, want_grads, want_dmem = DB.loss_and_grads(sd, cfg, memory, m.tokenizer.encode(meta['labels']), perms,
                                                                      O.attn_masks_from_perm, trace, DB.Dropout(p_drop, seed))
    report = []
    for name, want in trace.items():          # forward intermediates first: the first line that is off names the kernel
        got = res.intermediate(name, want.numel()).cpu().view(want.shape)
        err = float((got - want).abs().max())
        report.append(f'{name}: max|d| {err:.3e} of {float(want.abs().max()):.3e}')
    print('\n'.join(report))
    if p_drop:
        dropped = float((res.intermediate('hact', trace['hact'].numel()) == 0).float().mean())
        assert abs(dropped - p_drop) < 5e-3, dropped
        assert abs(float(want_loss) - meta['loss']) > 1e-3         # the masks do change the loss
    else:
        assert abs(float(res.loss) - meta['loss']) <= 1e-4 * meta['loss'], report
    assert abs(float(res.loss) - float(want_loss)) <= 1e-4 * float(want_loss), report
    assert (res.perm_losses.cpu() - want_pp).abs().max() <= 1e-4 * float(want_pp.max())
    bad = []
    for key, want in meta['grads'].items():
        got = res.grads[key].cpu()
        if key.startswith('encoder.'):
            assert not got.any()             # the decoder stage leaves the encoder's slots untouched
            continue
        ref = want_grads[key]
        err = float((got - ref).abs().max())
        tol = 1e-4 * max(float(ref.abs().max()), 1e-6) + 1e-7
        norm = float(got.double().norm())
        if err > tol or (not p_drop and abs(norm - want['norm']) > 1e-3 * max(want['norm'], 1e-6)):
            bad.append((key, err, tol, norm, want['norm']))
        if not p_drop and 'grad.' + key in g:               # the reference's own tensor
            assert (got - g['grad.' + key]).abs().max() <= tol, key
    assert not bad, (bad, report)
    err = float((res.dmemory.cpu() - want_dmem).abs().max())
    assert err <= 1e-4 * float(want_dmem.abs().max()), (err, report)


@pytest.mark.gpu
@pytest.mark.parametrize('p_drop', [0.0, 0.1])
def test_decoder_backward_bf16_operands_with_dropout(train_golden, p_drop):
    """`parseq_train_decoder` in the bf16-operand mode (every Linear product AND the attention products of
    train_attn_dec_bf16_kernel on bf16 operands), with and without dropout: the masks come from the same counter-based generator as
    the fp32 kernels', so the CPU backward with the SAME masks and exact fp32 arithmetic is the reference and the only difference
    is operand rounding — held to the budget test_bf16_operand_rounding_budget measured (per-tensor L2 error <= 6e-2, cosine >= 0.998,
    loss 5e-4), memory gradient included.  The encoder output is taken from the CPU oracle so that the comparison is the decoder's alone."""
    from gpu_util import DEV, make_model
    from oracle import decoder_backward as DB
    from parseq_amd.train import decoder_backward
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    m = make_model('parseq', 'fp32')
    m.train_precision = 'bf16'
    perms = g['perms'].long()
    seed = 0x0FEDCBA987654321
    with torch.no_grad():
        memory = O.encode(sd, cfg, g['images'])
        trace = {}
        want_loss, _, want_grads, want_dmem = DB.loss_and_grads(sd, cfg, memory, m.tokenizer.encode(meta['labels']), perms,
                                                                O.attn_masks_from_perm, trace, DB.Dropout(p_drop, seed))
    res = decoder_backward(m, g['images'].to(DEV), meta['labels'], perms, memory=memory.to(DEV), dropout=p_drop, seed=seed)
    torch.cuda.synchronize()
    assert abs(float(res.loss) - float(want_loss)) <= 5e-4 * float(want_loss)
    if p_drop:      # the same probabilities were dropped: the cross-attention output's zero pattern is the masks' (a dropped hidden unit is exactly 0)
        got_h = res.intermediate('hact', trace['hact'].numel()).cpu().view(trace['hact'].shape)
        assert torch.equal(got_h == 0, trace['hact'] == 0)
    rel, cos = [], []
    for key, ref in list(want_grads.items()) + [('d memory', want_dmem)]:
        if key.startswith('encoder.'):
            continue
        got = res.dmemory if key == 'd memory' else res.grads[key]
        a, b = ref.double().flatten(), got.cpu().double().flatten()
        if float(a.norm()) < 1e-7:
            continue
        rel.append((float((a - b).norm() / a.norm()), key))
        cos.append(float(a @ b / (a.norm() * b.norm())))
    rel.sort()
    print(f'bf16 decoder backward, dropout {p_drop}: per-tensor L2 error median {rel[len(rel) // 2][0]:.2e}, worst {rel[-1][0]:.2e} ({rel[-1][1]}), min cosine {min(cos):.5f}')
    assert 1e-4 < rel[len(rel) // 2][0] < 2e-2 and rel[-1][0] < 6e-2 and min(cos) > 0.998


@pytest.mark.gpu
def test_full_step_gradients_match_reference(train_golden):
    """Encoder forward (fp32, activations kept) -> decoder forward / backward -> encoder backward: the loss and the gradient of
    all 175 parameters against the reference's `training_step` + `loss.backward()` (tests/golden/parseq_train.*; every tensor
    by norm, fifteen whole), and every tensor against the hand-derived CPU backward."""
    from gpu_util import DEV, make_model
    from oracle import decoder_backward as DB, encoder_backward as EB
    from parseq_amd.train import loss_and_grads
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    m = make_model('parseq', 'bf16')                       # the step computes in fp32 whatever the inference precision is
    perms = g['perms'].long()
    res = loss_and_grads(m, g['images'].to(DEV), meta['labels'], perms)
    torch.cuda.synchronize()
    with torch.no_grad():
        memory, saved = EB.forward(sd, cfg, g['images'])
        _, _, want, dmem = DB.loss_and_grads(sd, cfg, memory, m.tokenizer.encode(meta['labels']), perms, O.attn_masks_from_perm)
        want.update(EB.backward(sd, cfg, saved, dmem))
    assert (res.memory.cpu() - memory).abs().max() <= 1e-4
    assert abs(float(res.loss) - meta['loss']) <= 1e-4 * meta['loss']
    assert set(res.grads) == set(meta['grads']) and len(res.grads) == 175
    bad = []
    for key, ref in meta['grads'].items():
        got = res.grads[key].cpu()
        err = float((got - want[key]).abs().max())
        tol = 2e-4 * max(float(want[key].abs().max()), 1e-6) + 1e-7
        norm = float(got.double().norm())
        if err > tol or abs(norm - ref['norm']) > 1e-3 * max(ref['norm'], 1e-6):
            bad.append((key, err, tol, norm, ref['norm']))
        if 'grad.' + key in g:
            assert (got - g['grad.' + key]).abs().max() <= tol, key
    assert not bad, bad


@pytest.mark.gpu
def test_three_optimiser_steps_follow_torch_adamw(train_golden):
    """TrainStep (forward, backward, gradient-norm clipping, AdamW under the OneCycle schedule, all on the device) against the
    same three steps on the CPU: autograd through the oracle, torch.nn.utils.clip_grad_norm_, torch.optim.AdamW + OneCycleLR.
    Clipping is made active (max norm 5 < the 11.1 of the first gradient).  Afterwards the module's own tensors and the
    inference plans carry the updated weights."""
    from gpu_util import DEV, make_model
    from parseq_amd.train import TrainStep
    g, meta = train_golden
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    perms = g['perms'].long()
    images = g['images']
    step = TrainStep(m, total_steps=40, clip_val=5.0, weight_decay=0.01)
    lrs, got_losses = [], []
    for _ in range(3):
        lrs.append(step.lr)
        got_losses.append(float(step(images.to(DEV), meta['labels'], perms)))
    torch.cuda.synchronize()
    # CPU reference
    sd = {k: v.clone().requires_grad_(True) for k, v in synth_state_dict(cfg, 0).items()}
    decay = [v for k, v in sd.items() if v.ndim > 1 and not k.endswith('.bias')]
    rest = [v for k, v in sd.items() if not (v.ndim > 1 and not k.endswith('.bias'))]
    opt = torch.optim.AdamW([{'params': decay, 'weight_decay': 0.01}, {'params': rest, 'weight_decay': 0.0}], lr=step.max_lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, step.max_lr, 40, pct_start=m.warmup_pct, cycle_momentum=False)
    tgt = m.tokenizer.encode(meta['labels'])
    want_losses = []
    for i in range(3):
        assert abs(opt.param_groups[0]['lr'] - lrs[i]) <= 1e-9 * step.max_lr
        opt.zero_grad()
        loss = O.training_loss(sd, cfg, images, tgt, perms)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(sd.values()), 5.0)
        if i == 0:
            first_grads = {k: v.grad.clone() for k, v in sd.items()}
        opt.step()
        sched.step()
        want_losses.append(float(loss.detach()))
    assert all(abs(a - b) <= 2e-4 * b for a, b in zip(got_losses, want_losses)), (got_losses, want_losses)
    assert got_losses[2] < got_losses[0]
    # Parameters: Adam divides by sqrt(v), so where the true gradient is zero up to round-off (the key bias of every attention:
    # soft-max is invariant to it) the update is +-lr times noise on both sides; compare where the first gradient is above that
    # floor, and bound the rest by Adam's own step bound.
    start = synth_state_dict(cfg, 0)
    lr_sum = sum(lrs)
    bad = []
    for key, t in m.model.state_dict().items():
        d_got, d_want = t.cpu() - start[key], sd[key].detach() - start[key]
        err = (d_got - d_want).abs()
        real = first_grads[key].abs() > 1e-6
        if real.any():
            if float(err[real].max()) > 0.1 * lr_sum or float(err[real].mean()) > 2e-3 * lr_sum:
                bad.append((key, 'real', float(err[real].max()), float(err[real].mean())))
        if float(d_got.abs().max()) > 1.05 * lr_sum + 0.02 * float(start[key].abs().max()) * lr_sum:
            bad.append((key, 'bound', float(d_got.abs().max())))
        if float(d_want.abs().max()) > 0 and float(d_got.abs().max()) == 0:
            bad.append((key, 'unchanged'))
    assert not bad, (bad, lr_sum)
    # the inference path now runs on the updated weights (plans were re-packed): fp32-mode logits against the oracle with them
    m.precision = 'fp32'
    with torch.inference_mode():
        got = m(images.to(DEV), 25).float().cpu()
        want = O.forward({k: v.detach() for k, v in sd.items()}, cfg, images, 25, decode_ar=True, refine_iters=1)
    assert (got - want).abs().max() <= 2e-3, float((got - want).abs().max())


@pytest.mark.gpu
def test_batch_64_step_uses_the_matrix_core_gemms():
    """64 crops x 26 positions = 1664 decoder rows and 8192 encoder rows: every Linear (forward, dX, dW with split-K) runs on the
    MFMA fp32 GEMM and the column sums take their two-stage form.  All 175 gradients against the hand-derived CPU backward."""
    from gpu_util import DEV, make_model
    from oracle import decoder_backward as DB, encoder_backward as EB
    from parseq_amd.train import loss_and_grads
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    m = make_model('parseq', 'bf16')
    gen = torch.Generator().manual_seed(99)
    images = synth_images(64, cfg, seed=77)
    lengths = torch.randint(1, 26, (64,), generator=gen).tolist()
    lengths[5] = 25
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]
    m.rng = np.random.default_rng(8)
    torch.manual_seed(9)
    res = loss_and_grads(m, images.to(DEV), labels)
    torch.cuda.synchronize()
    assert res.perms.shape == (6, 27)
    with torch.no_grad():
        memory, saved = EB.forward(sd, cfg, images)
        want_loss, _, want, dmem = DB.loss_and_grads(sd, cfg, memory, m.tokenizer.encode(labels), res.perms, O.attn_masks_from_perm)
        want.update(EB.backward(sd, cfg, saved, dmem))
    assert abs(float(res.loss) - float(want_loss)) <= 1e-4 * float(want_loss)
    bad = []
    for key, ref in want.items():
        got = res.grads[key].cpu()
        err = float((got - ref).abs().max())
        tol = 3e-4 * max(float(ref.abs().max()), 1e-6) + 1e-7
        if err > tol:
            bad.append((key, err, tol))
    assert not bad, bad


@pytest.mark.gpu
def test_large_batch_step_is_the_weighted_sum_of_its_halves():
    """704 crops with a 25-character label: 6 x 704 x 26 = 109 824 decoder rows and 90 112 encoder rows per LayerNorm backward — more than
    the fixed 16 M-float scratch held as partial rows (the step used to abort there); the layouts now size the scratch from the
    row count.  Property that needs no oracle at this size: with the permutations given and dropout off, the loss is a mean over the
    non-<pad> targets, so the whole batch's gradient is the target-count-weighted sum of its two halves' gradients."""
    from gpu_util import DEV, make_model
    from parseq_amd.train import loss_and_grads, loss_denominator
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    m.train_precision = 'bf16'
    gen = torch.Generator().manual_seed(5)
    B = 704
    images = synth_images(B, cfg, seed=31).to(DEV)
    lengths = torch.randint(1, 26, (B,), generator=gen).tolist()
    lengths[3] = lengths[B // 2 + 3] = 25                      # both halves see L = 26
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]
    m.rng = np.random.default_rng(8)
    torch.manual_seed(9)
    whole = loss_and_grads(m, images, labels, dropout=0.0)
    torch.cuda.synchronize()
    assert torch.isfinite(whole.loss)
    halves = [loss_and_grads(m, images[i:i + B // 2], labels[i:i + B // 2], perms=whole.perms, dropout=0.0) for i in (0, B // 2)]
    n = [loss_denominator(labels[i:i + B // 2], whole.perms.shape[0]) for i in (0, B // 2)]
    want_loss = (n[0] * float(halves[0].loss) + n[1] * float(halves[1].loss)) / (n[0] + n[1])
    assert abs(float(whole.loss) - want_loss) <= 2e-5 * want_loss
    want = (n[0] * halves[0].flat.double() + n[1] * halves[1].flat.double()) / (n[0] + n[1])
    got = whole.flat.double()
    rel = float((got - want).norm() / want.norm())
    assert rel <= 2e-3, rel                                     # bf16-operand mode: only the dW contractions' summation order differs
    for key in ('encoder.norm.weight', 'encoder.blocks.0.norm1.bias', 'decoder.norm.weight', 'decoder.layers.0.norm_c.bias'):
        g, w = whole.grads[key].double(), (n[0] * halves[0].grads[key].double() + n[1] * halves[1].grads[key].double()) / (n[0] + n[1])
        assert float((g - w).abs().max()) <= 2e-3 * float(w.abs().max()) + 1e-9, key


@pytest.mark.gpu
def test_overlapped_allreduce_step_equals_the_plain_step():
    """The data-parallel step with the gradient all-reduce riding behind the backward's events (TrainStep.overlap_allreduce: one RCCL
    all-reduce per gradient segment on a side stream, started by the event parseq_train_encoder_backward records when that segment is
    final — DDP's bucketed reducer, reference train.py:65-71) against the same step with no collective at all, on the one GPU a test box
    has: a one-rank RCCL group with the collectives forced.  One rank's all-reduce is the identity, so loss and every updated weight must
    agree bit for bit after two steps — which they only do if every segment was reduced after its last writer and before the optimiser
    read it.  Also: the segments tile the flat buffer in the documented completion order and carry live events."""
    import socket

    import torch.distributed as dist
    from gpu_util import DEV, make_model
    from parseq_amd import _native
    from parseq_amd.train import TrainStep, grad_segments
    cfg = CONFIGS['parseq']
    gen = torch.Generator().manual_seed(3)
    B = 48
    images = synth_images(B, cfg, seed=21).to(DEV)
    lengths = torch.randint(1, 26, (B,), generator=gen).tolist()
    lengths[0] = 25
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        runs = []
        for overlapped in (True, False):
            m = make_model('parseq', 'bf16')
            m.train()
            m.train_precision = 'bf16'
            m.rng = np.random.default_rng(11)
            torch.manual_seed(12)
            step = TrainStep(m, total_steps=10)
            step.overlap_allreduce, step.force_collectives = overlapped, overlapped
            losses = [float(step(images, labels)) for _ in range(2)]
            torch.cuda.synchronize()
            runs.append((losses, {k: v.detach().clone() for k, v in m.model.state_dict().items()}, step))
        (la, wa, sa), (lb, wb, _) = runs
        assert la == lb, (la, lb)
        assert all(torch.equal(wa[k], wb[k]) for k in wa), [k for k in wa if not torch.equal(wa[k], wb[k])][:5]
        assert sa._comm_stream is not None                       # the overlapped path really ran
        native = sa.system.model._sync_native().model
        segs = grad_segments(native)
        n = _native.lib().parseq_model_grad_elems(native)
        assert len(segs) == cfg.enc_depth + 1 and all(ev for _, _, ev in segs)
        assert segs[0][1] == n and segs[-1][0] == 0               # the decoder's part first, the head of the buffer last
        assert [b for b, _, _ in segs] == sorted((b for b, _, _ in segs), reverse=True)      # back to front
        assert all(a[0] == b[1] for a, b in zip(segs, segs[1:]))  # contiguous: they tile [0, n)
        # the events belong to the step that recorded them: once the next step's decoder has started writing gradients, and until an encoder backward has run to its
        # end again, asking for a segment's event is a state error (it used to hand out the previous step's event, already signalled)
        lib = _native.lib()
        d = torch.zeros(64, device=DEV)
        dp = _native.ptr(d)
        # (a decoder call that gets as far as its shape check — ctx_len 1 is refused — has already declared the new step)
        assert lib.parseq_train_decoder(native, dp, dp, dp, dp, dp, 1, 1, 1, 1, 0.0, 0, dp, dp, dp, dp, 256, _native.stream_ptr()) != 0
        b, e, ev = C.c_int64(), C.c_int64(), C.c_void_p()
        assert lib.parseq_train_grad_segment(native, 0, C.byref(b), C.byref(e), C.byref(ev)) != 0
        assert lib.parseq_train_grad_segment(native, 0, C.byref(b), C.byref(e), None) == 0      # the ranges alone stay available
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bf16_operand_step_within_the_rounding_budget():
    """system.train_precision = 'bf16' (parseq_model_set_train_precision: both operands of every aligned Linear product rounded to
    bfloat16, fp32 accumulate, fp32 everything else — train_ops.h mfma_bgemm_kernel) on the batch-64 step whose Linears all take the
    matrix-core path: loss and all 175 gradients against the EXACT fp32 CPU backward, within the budget
    test_bf16_operand_rounding_budget measured for this rounding (loss 5e-4 relative, per-tensor L2 error <= 6e-2, cosine >= 0.998) —
    and visibly not the fp32 path (median per-tensor error above 1e-4)."""
    from gpu_util import DEV, make_model
    from oracle import decoder_backward as DB, encoder_backward as EB
    from parseq_amd.train import loss_and_grads
    cfg = CONFIGS['parseq']
    sd = synth_state_dict(cfg, 0)
    m = make_model('parseq', 'bf16')
    m.train_precision = 'bf16'
    gen = torch.Generator().manual_seed(99)
    images = synth_images(64, cfg, seed=77)
    lengths = torch.randint(1, 26, (64,), generator=gen).tolist()
    lengths[5] = 25
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]
    m.rng = np.random.default_rng(8)
    torch.manual_seed(9)
    res = loss_and_grads(m, images.to(DEV), labels)
    torch.cuda.synchronize()
    with torch.no_grad():
        memory, saved = EB.forward(sd, cfg, images)
        want_loss, _, want, dmem = DB.loss_and_grads(sd, cfg, memory, m.tokenizer.encode(labels), res.perms, O.attn_masks_from_perm)
        want.update(EB.backward(sd, cfg, saved, dmem))
    assert abs(float(res.loss) - float(want_loss)) <= 5e-4 * float(want_loss)
    rel, cos = [], []
    for key, ref in want.items():
        a, b = ref.double().flatten(), res.grads[key].cpu().double().flatten()
        if float(a.norm()) < 1e-7:
            continue
        rel.append((float((a - b).norm() / a.norm()), key))
        cos.append(float(a @ b / (a.norm() * b.norm())))
    rel.sort()
    print(f'bf16-operand step: per-tensor L2 error median {rel[len(rel) // 2][0]:.2e}, worst {rel[-1][0]:.2e} ({rel[-1][1]}), min cosine {min(cos):.5f}')
    assert 1e-4 < rel[len(rel) // 2][0] < 2e-2 and rel[-1][0] < 6e-2 and min(cos) > 0.998
    # the switch is per model and reversible: back in fp32 the same call meets the fp32 gate again
    m.train_precision = 'fp32'
    m.rng = np.random.default_rng(8)
    torch.manual_seed(9)
    res32 = loss_and_grads(m, images.to(DEV), labels, res.perms)
    key = 'encoder.blocks.0.mlp.fc1.weight'
    assert float((res32.grads[key].cpu() - want[key]).abs().max()) <= 3e-4 * float(want[key].abs().max()) + 1e-7


@pytest.mark.gpu
def test_bf16_shadow_operands_change_no_bit():
    """Round 3: in the bf16-operand mode the encoder's activations that only feed Linear products (LayerNorm outputs, attention output,
    GELU output), the Linear weights and the gradients on the dX side are WRITTEN as bfloat16 by their producers and read by the
    64-deep matrix-core GEMM (train_ops.h mfma_bgemm16_kernel / the B16 forms of mfma_bgemm_kernel).  That is the same
    round-to-nearest-even the fp32-in-memory path applies on the way into LDS, in the same accumulation order: loss and every one of
    the 175 gradients must be bit-identical with PARSEQ_TRAIN_NO_SHADOWS=1 (every operand fp32 in memory)."""
    import os
    from gpu_util import DEV, make_model
    from parseq_amd.train import loss_and_grads
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    m.train_precision = 'bf16'
    gen = torch.Generator().manual_seed(5)
    images = synth_images(32, cfg, seed=21).to(DEV)
    lengths = torch.randint(1, 26, (32,), generator=gen).tolist()
    lengths[3] = 25
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]

    def run():
        m.rng = np.random.default_rng(8)
        torch.manual_seed(9)
        r = loss_and_grads(m, images, labels)
        torch.cuda.synchronize()
        return float(r.loss), {k: v.clone() for k, v in r.grads.items()}

    assert 'PARSEQ_TRAIN_NO_SHADOWS' not in os.environ and 'PARSEQ_TRAIN_SHADOW_LEVEL' not in os.environ
    os.environ['PARSEQ_TRAIN_SHADOW_LEVEL'] = '1'      # shadows BESIDE the fp32 copies; the default (bf16-only storage of the fc1 pre-activation
    try:                                               # and of two gradients) is a different rounding, gated by the oracle tests above
        loss_a, grads_a = run()
    finally:
        del os.environ['PARSEQ_TRAIN_SHADOW_LEVEL']
    os.environ['PARSEQ_TRAIN_NO_SHADOWS'] = '1'
    try:
        loss_b, grads_b = run()
    finally:
        del os.environ['PARSEQ_TRAIN_NO_SHADOWS']
    assert loss_a == loss_b
    diff = [k for k in grads_a if not torch.equal(grads_a[k], grads_b[k])]
    assert not diff, diff


@pytest.mark.gpu
def test_one_launch_training_forward_against_the_per_operation_forward():
    """Round 5: in the bf16-operand mode the encoder's training forward is ONE launch (encoder_blocks.h record mode: the inference
    throughput kernel's walk over the twelve blocks with the residual rows resident, writing the record the hand-derived backward reads —
    x, LayerNorm outputs, q | k | v, attention output, fc1 pre-activation and its GELU) instead of 48 GEMM + 24 LayerNorm + 12 attention
    launches.  Same operands rounded to bfloat16, fp32 accumulation; the rounding POINTS differ a little (probabilities rounded before
    the normalisation, the polynomial GELU of common.h), so the step is compared with the per-operation forward
    (PARSEQ_TRAIN_ENC_PER_OP=1) at the bf16 noise level — loss 2e-3 relative, every gradient within 5e-2 in L2 and cosine 0.999 — and
    must not be that path (some gradient differs).  The gate against the exact fp32 backward is
    test_bf16_operand_step_within_the_rounding_budget, which runs through the one-launch forward."""
    import os
    from gpu_util import DEV, make_model
    from parseq_amd.train import loss_and_grads
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    m.train_precision = 'bf16'
    gen = torch.Generator().manual_seed(15)
    images = synth_images(40, cfg, seed=23).to(DEV)
    lengths = torch.randint(1, 26, (40,), generator=gen).tolist()
    lengths[3] = 25
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]

    def run():
        m.rng = np.random.default_rng(8)
        torch.manual_seed(9)
        r = loss_and_grads(m, images, labels)
        torch.cuda.synchronize()
        return float(r.loss), {k: v.clone() for k, v in r.grads.items()}

    assert 'PARSEQ_TRAIN_ENC_PER_OP' not in os.environ
    loss_a, grads_a = run()
    os.environ['PARSEQ_TRAIN_ENC_PER_OP'] = '1'
    try:
        loss_b, grads_b = run()
    finally:
        del os.environ['PARSEQ_TRAIN_ENC_PER_OP']
    assert abs(loss_a - loss_b) <= 2e-3 * abs(loss_b), (loss_a, loss_b)
    rel, cos = [], []
    for k, b in grads_b.items():
        a, b = grads_a[k].double().flatten(), b.double().flatten()
        if float(b.norm()) < 1e-7:
            continue
        rel.append((float((a - b).norm() / b.norm()), k))
        cos.append(float(a @ b / (a.norm() * b.norm())))
    rel.sort()
    print(f'one-launch vs per-operation forward: loss {loss_a:.6f} / {loss_b:.6f}, per-tensor L2 difference median {rel[len(rel) // 2][0]:.2e}, '
          f'worst {rel[-1][0]:.2e} ({rel[-1][1]}), min cosine {min(cos):.5f}')
    assert rel[-1][0] < 5e-2 and min(cos) > 0.999
    assert any(not torch.equal(grads_a[k], grads_b[k]) for k in grads_a)


@pytest.mark.gpu
def test_two_stream_backward_changes_no_bit():
    """Round 5: in the bf16-operand mode parseq_train_encoder_backward runs the blocks' weight-gradient products on a second stream beside
    the dX / LayerNorm / attention chain (events order the two; lib_train.hip).  Same kernels on the same operands in the same order per
    buffer: loss and every gradient must equal the one-stream schedule (PARSEQ_TRAIN_ONE_STREAM=1) bit for bit — which they only do if no
    product ever read a gradient buffer before its producer finished or after its next writer started.  Three runs of each, batch 48."""
    import os
    from gpu_util import DEV, make_model
    from parseq_amd.train import loss_and_grads
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    m.train_precision = 'bf16'
    gen = torch.Generator().manual_seed(25)
    images = synth_images(48, cfg, seed=27).to(DEV)
    lengths = torch.randint(1, 26, (48,), generator=gen).tolist()
    lengths[3] = 25
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]

    def run():
        m.rng = np.random.default_rng(8)
        torch.manual_seed(9)
        r = loss_and_grads(m, images, labels)
        torch.cuda.synchronize()
        return float(r.loss), {k: v.clone() for k, v in r.grads.items()}

    assert 'PARSEQ_TRAIN_ONE_STREAM' not in os.environ
    two = [run() for _ in range(3)]
    os.environ['PARSEQ_TRAIN_ONE_STREAM'] = '1'
    try:
        one = run()
    finally:
        del os.environ['PARSEQ_TRAIN_ONE_STREAM']
    for loss, grads in two:
        assert loss == one[0]
        diff = [k for k in grads if not torch.equal(grads[k], one[1][k])]
        assert not diff, diff[:5]


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 3, 130])
def test_one_launch_training_forward_at_ragged_batch_sizes(batch):
    """parseq_train_encoder_forward through the C ABI, one launch against per-operation launches, at batch sizes that leave most of the chip
    empty (1, 3) and that need a second, partly filled round of workgroups (130 images on 256 CUs is one round; the point is a grid that is
    not a multiple of anything): `memory` within the bf16 noise of the two forwards (3e-2 on values of about 4)."""
    import os
    from gpu_util import DEV, make_model
    from parseq_amd import _native
    from parseq_amd.train import _set_train_precision
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    m.train_precision = 'bf16'
    lib = _native.lib()
    native = m.model._sync_native().model
    _set_train_precision(m, native)
    images = synth_images(batch, cfg, seed=31).to(DEV)
    nbytes = lib.parseq_train_encoder_workspace_bytes(native, batch)

    def fwd():
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=DEV)
        mem = torch.empty(batch, 128, 384, dtype=torch.float32, device=DEV)
        _native.check(lib.parseq_train_encoder_forward(native, _native.ptr(images), batch, _native.ptr(mem), _native.ptr(ws), nbytes, _native.stream_ptr(images)))
        torch.cuda.synchronize()
        return mem.cpu()

    assert 'PARSEQ_TRAIN_ENC_PER_OP' not in os.environ
    a = fwd()
    os.environ['PARSEQ_TRAIN_ENC_PER_OP'] = '1'
    try:
        b = fwd()
    finally:
        del os.environ['PARSEQ_TRAIN_ENC_PER_OP']
    assert torch.isfinite(a).all()
    assert float((a - b).abs().max()) <= 3e-2 and not torch.equal(a, b), float((a - b).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_permutation_passes_as_one_batch_equal_one_after_the_other(train_golden, precision, monkeypatch):
    """Round 3: the decoder runs the K permutation passes of a step as ONE batch of K * B images (parseq_train_decoder,
    PARSEQ_TRAIN_PERM_GROUP).  Against the passes one after the other (group 1, the arrangement every earlier gate was measured on), with
    dropout on: the SAME masks (the zero pattern of the dropped hidden units of the last pass is identical), the same per-pass losses
    and every decoder-side gradient and d loss / d memory equal up to the summation order of fp32 (and, in the bf16-operand mode, the
    operand roundings that order can flip).  Also the uneven grouping 4 + 2 and, in the bf16 mode, the cross-attention with and without
    the in-kernel walk over the passes (TrainAttnArgs::pass_loop)."""
    from gpu_util import DEV, make_model
    from parseq_amd.train import decoder_backward
    g, meta = train_golden
    m = make_model('parseq', 'fp32')
    m.train_precision = precision
    perms = g['perms'].long()
    assert len(perms) == 6
    with torch.no_grad():
        memory = O.encode(synth_state_dict(CONFIGS['parseq'], 0), CONFIGS['parseq'], g['images']).to(DEV)
    seed = 0x00C0FFEE12345678

    def run(group, no_loop=False):
        if group is None:
            monkeypatch.delenv('PARSEQ_TRAIN_PERM_GROUP', raising=False)
        else:
            monkeypatch.setenv('PARSEQ_TRAIN_PERM_GROUP', str(group))
        if no_loop:
            monkeypatch.setenv('PARSEQ_TRAIN_NO_PASS_LOOP', '1')
        else:
            monkeypatch.delenv('PARSEQ_TRAIN_NO_PASS_LOOP', raising=False)
        r = decoder_backward(m, g['images'].to(DEV), meta['labels'], perms, memory=memory, dropout=0.1, seed=seed)
        torch.cuda.synchronize()
        B, L, K = r._shape
        hact = r.intermediate('hact', B * L * 4 * 384).clone()
        out = (float(r.loss), r.perm_losses.clone(), {k: v.clone() for k, v in r.grads.items()}, r.dmemory.clone(), hact)
        del r
        return out

    base = run(1)
    tol = 2e-5 if precision == 'fp32' else 5e-3
    variants = [('all at once', run(None)), ('4 + 2', run(4))]
    if precision == 'bf16':
        variants.append(('all at once, per-pass d K | d V', run(None, no_loop=True)))
    for what, (loss, pp, grads, dmem, hact) in variants:
        assert torch.equal(hact == 0, base[4] == 0), what
        assert abs(loss - base[0]) <= tol * abs(base[0]), (what, loss, base[0])
        assert (pp - base[1]).abs().max() <= tol * float(base[1].abs().max()), what
        worst = []
        for key, ref in list(base[2].items()) + [('d memory', base[3])]:
            got = dmem if key == 'd memory' else grads[key]
            a, b = ref.double().flatten(), got.double().flatten()
            if float(a.norm()) < 1e-9:
                assert float(b.norm()) < 1e-9, (what, key)
                continue
            worst.append((float((a - b).norm() / a.norm()), key))
        worst.sort()
        print(f'{precision}, {what}: worst per-tensor L2 difference {worst[-1][0]:.2e} ({worst[-1][1]})')
        assert worst[-1][0] <= tol, (what, worst[-3:])


@pytest.mark.gpu
def test_optimiser_step_hands_every_weight_back_in_one_launch(train_golden):
    """`parseq_model_get_params` (one launch over a table of copy pieces) leaves in the module's tensors exactly what the per-tensor
    `parseq_model_get_param` copies do."""
    import ctypes as C
    from gpu_util import DEV, make_model
    from parseq_amd import _native
    from parseq_amd.train import TrainStep
    g, meta = train_golden
    m = make_model('parseq', 'fp32')
    step = TrainStep(m, total_steps=10)
    step(g['images'].to(DEV), meta['labels'], g['perms'].long())
    torch.cuda.synchronize()
    lib, st = _native.lib(), m.model._native_state
    for key, t in m.model.state_dict().items():
        one = torch.empty_like(t)
        _native.check(lib.parseq_model_get_param(st.model, key.encode(), _native.ptr(one), one.numel(), _native.stream_ptr(one)))
        torch.cuda.synchronize()
        assert torch.equal(one, t), key


@pytest.mark.gpu
def test_one_launch_training_forward_record_slot_by_slot(monkeypatch):
    """the record mode's ring-stage waits let the record's own stores stay in flight (encoder_blocks.h kRecStores*); a wait that
    became too loose would let a stage be read before it has landed, and `memory` alone (bf16-noise tolerance) might not show it.  So the
    RECORD is compared slot by slot — x, q | k | v, attention output, x after the attention branch, fc1 pre-activation, its GELU, the two
    LayerNorm outputs, for each of the twelve blocks — between the one launch and the per-operation forward (PARSEQ_TRAIN_ENC_PER_OP=1),
    batch 8: relative rms within bf16 rounding compounded over the blocks (measured 2e-5 in block 0 growing to 5e-3 in block 11).  The
    layout is lib_train.hip TrainEncoderLayout's, restated here (a change there fails this test loudly, which is the point)."""
    from gpu_util import DEV, make_model
    from parseq_amd import _native
    from parseq_amd.train import _set_train_precision
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    m.train_precision = 'bf16'
    B, E, F, S, PK, depth = 8, 384, 1536, 128, 96, 12
    images = synth_images(B, cfg, seed=23).to(DEV)
    lib = _native.lib()
    native = m.model._sync_native().model
    _set_train_precision(m, native)
    MS = B * S
    r64 = lambda n: (n + 63) // 64 * 64
    layer0, cur, slots = r64(MS * PK), 0, {}
    for name, n in [('x', MS * E), ('qkv', MS * 3 * E), ('ao', MS * E), ('x_mid', MS * E), ('hpre', MS * F), ('hact', MS * F), ('n1', MS * E), ('n2', MS * E)]:
        slots[name] = (cur, n)
        cur += r64(n)
    stride = cur
    nbytes = lib.parseq_train_encoder_workspace_bytes(native, B)
    assert nbytes // 4 >= layer0 + stride * depth + MS * E

    def fwd():
        ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=DEV)
        mem = torch.empty(B, S, E, dtype=torch.float32, device=DEV)
        _native.check(lib.parseq_train_encoder_forward(native, _native.ptr(images), B, _native.ptr(mem), _native.ptr(ws), nbytes, _native.stream_ptr(images)))
        torch.cuda.synchronize()
        return ws, mem
    monkeypatch.delenv('PARSEQ_TRAIN_ENC_PER_OP', raising=False)
    wa, ma = fwd()
    monkeypatch.setenv('PARSEQ_TRAIN_ENC_PER_OP', '1')
    wb, mb = fwd()
    assert not torch.equal(wa, wb)                                   # two different forwards ran

    def view(ws, layer, name):
        o, n = slots[name]
        t = ws[layer0 + layer * stride + o: layer0 + layer * stride + o + n]
        return t.view(torch.bfloat16)[:n].float() if name in ('ao', 'hpre', 'hact', 'n1', 'n2') else t
    worst = {}
    for layer in range(depth):
        for name in slots:
            a, b = view(wa, layer, name), view(wb, layer, name)
            assert torch.isfinite(a).all() and torch.isfinite(b).all(), (layer, name)
            rel = float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12))
            worst[name] = max(worst.get(name, 0.0), rel)
            # block 0: the two forwards' values agree to fp32 rounding BEFORE storage; a slot stored as bf16 then differs where a value sits on a
            # rounding boundary (relative rms of one bf16 ulp flip ~ 2^-9: measured 1.0e-3 on hpre), a fp32 slot hardly at all
            first = 4e-3 if name in ('ao', 'hpre', 'hact', 'n1', 'n2') else 1e-3
            assert rel <= (first if layer == 0 else 2e-2), (layer, name, rel)
    print('[record, one launch vs per-operation] worst relative rms per slot over the twelve blocks:', {k: f'{v:.1e}' for k, v in worst.items()})
    rel = float((ma - mb).pow(2).mean().sqrt() / mb.pow(2).mean().sqrt())
    assert rel <= 2e-2, rel


@pytest.mark.gpu
@pytest.mark.parametrize('parts', [2, 4])
def test_micro_batched_step_equals_the_one_piece_step(parts):
    """Round 6: a step cut into micro-batches that run AT ONCE on separate streams (loss_and_grads_micro: own workspaces and gradient
    buffers per part, the library's second backward stream per caller stream, one tokenisation / one draw of permutations / the batch's
    loss denominator shared) must produce the one-piece step's loss and gradients — dropout off: the same products, only the fp32 summation
    order of the weight gradients differs.  Run three times: the parts race each other on the device, a shared buffer would show."""
    from gpu_util import DEV, make_model
    from parseq_amd.train import loss_and_grads, loss_and_grads_micro
    cfg = CONFIGS['parseq']
    m = make_model('parseq', 'bf16')
    m.train_precision = 'fp32'                            # exact products: what is left between the two schedules is fp32 summation order
    gen = torch.Generator().manual_seed(31)
    B = 64
    images = synth_images(B, cfg, seed=29).to(DEV)
    lengths = torch.randint(1, 26, (B,), generator=gen).tolist()
    lengths[B - 3] = 25                                   # the longest label sits in the LAST part: every part must still run 26 positions
    labels = [''.join(CHARSET_94[int(i)] for i in torch.randint(0, 94, (n,), generator=gen)) for n in lengths]
    m.rng = np.random.default_rng(4)
    torch.manual_seed(5)
    perms = m.gen_tgt_perms(m.tokenizer.encode(labels))
    one = loss_and_grads(m, images, labels, perms)
    torch.cuda.synchronize()
    want_loss, want = float(one.loss), one.flat.clone()
    for _ in range(3):
        got = loss_and_grads_micro(m, images, labels, perms, parts=parts)
        torch.cuda.synchronize()
        assert abs(float(got.loss) - want_loss) <= 5e-6 * abs(want_loss), (float(got.loss), want_loss)
        d = (got.flat - want).double()
        rel = float(d.norm() / want.double().norm())
        assert rel <= 1e-5, rel
        for k, g in got.grads.items():                    # per tensor as well: a small tensor's error would drown in the norm of the whole buffer
            w = one.grads[k].double()
            if float(w.norm()) > 1e-9:
                assert float((g.double() - w).norm() / w.norm()) <= 1e-4, k
