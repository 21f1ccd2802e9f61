"""GPU parity tests of the whole hot path (hubconf-level model -> libparseq_hip) against

  * the golden vectors minted from the REFERENCE's own model code (tests/golden, oracle/make_golden.py), and
  * the CPU oracle (oracle/parseq_oracle.py) on the same seeded inputs, in both arithmetic modes.

Bars (BASELINE.json north_star; SURVEY.md section 7 hard part 1):
  fp32 mode : |dlogit| <= 1e-3 against the reference outputs, argmax- and string-identical, every decode mode.
  bf16 mode : per-kernel tests (test_hip_ops.py) are the sharp check — identical bf16 operands, fp32-accumulate error only.
              End to end the comparison is against the rounding-aware oracle (same bf16 rounding points): |dlogit| <= 3e-2
              (measured 1.3e-2..1.7e-2; two bf16 evaluations decorrelate after a few layers because a 1e-7 fp32 difference
              that flips one bf16 rounding is a 4e-3 perturbation downstream), and against the exact fp32 reference
              |dlogit| <= 6e-2 (measured 2.2e-2..3.1e-2) with argmax identity required on every position whose reference
              top-1/top-2 margin exceeds 2x that bound (AR modes: up to the first position that does not).  The 1e-3 bar of
              the north star is met by the fp32 mode only; bf16 operands cannot meet it on these weights.
"""
import pytest
import torch

from gpu_util import DEV, make_model, report
from oracle import parseq_oracle as O
from oracle.synth import CONFIGS, synth_images, synth_state_dict

pytestmark = pytest.mark.gpu

MODES = {'nar0': (False, 0, None), 'nar1': (False, 1, None), 'ar0': (True, 0, None), 'ar0_full': (True, 0, 25),
         'ar0_len7': (True, 0, 7), 'ar1': (True, 1, None), 'ar2': (True, 2, None)}


@pytest.fixture(scope='module', params=['parseq', 'parseq-tiny'])
def name(request):
    return request.param


@pytest.fixture(scope='module')
def models(name):
    return {p: make_model(name, p) for p in ('fp32', 'bf16', 'bf16x3')}


def _run(m, images, mode):
    ar, ri, ml = MODES[mode]
    m.model.decode_ar, m.model.refine_iters = ar, ri
    with torch.inference_mode():
        out = m(images, ml)
    torch.cuda.synchronize()
    return out.float().cpu()


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_encoder_memory_fp32(name, models, golden, precision):
    g, _ = golden(name)
    mem = models[precision].model.encode(g['images'].to(DEV)).cpu()
    err, msg = report(f'{name} memory {precision} vs reference', mem, g['memory'])
    # fp32: 12 layers of fp32 reassociation on O(1) activations.  bf16x3: operands carry 16 mantissa bits (2^-17 per product)
    assert err <= (2e-4 if precision == 'fp32' else 5e-4), msg


def test_encoder_memory_bf16_vs_rounding_oracle(name, models, golden):
    g, _ = golden(name)
    cfg, sd = CONFIGS[name], synth_state_dict(CONFIGS[name], 0)
    mem = models['bf16'].model.encode(g['images'].to(DEV)).cpu()
    with torch.inference_mode():
        want = O.encode(sd, cfg, g['images'], rounding='bf16')
    err, msg = report(f'{name} memory bf16 vs rounding-aware oracle', mem, want)
    report(f'{name} memory bf16 vs exact fp32 reference (informative)', mem, g['memory'])
    assert err <= 3e-2, msg       # same rounding points; residual = fp32 reassociation amplified through bf16 re-rounding


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])     # the two modes that meet the north star's 1e-3 (bf16x3: at matrix-core speed)
@pytest.mark.parametrize('mode', list(MODES))
def test_forward_fp32_matches_reference(name, models, golden, mode, precision):
    g, meta = golden(name)
    m = models[precision]
    got = _run(m, g['images'].to(DEV), mode)
    ref = g[f'logits.{mode}']
    assert list(got.shape) == list(ref.shape) == meta['modes'][mode]['shape']
    err, msg = report(f'{name} {mode} {precision} logits vs reference', got, ref)
    assert err <= 1e-3, msg
    assert torch.equal(got.argmax(-1), ref.argmax(-1))
    strings, _ = m.tokenizer.decode(got.softmax(-1))
    assert strings == meta['modes'][mode]['strings']


@pytest.mark.parametrize('mode', ['nar0', 'ar0', 'ar1', 'ar2'])
def test_forward_bf16(name, models, golden, mode):
    g, meta = golden(name)
    cfg, sd = CONFIGS[name], synth_state_dict(CONFIGS[name], 0)
    m = models['bf16']
    ar, ri, ml = MODES[mode]
    got = _run(m, g['images'].to(DEV), mode)
    with torch.inference_mode():
        want = O.forward(sd, cfg, g['images'], ml, decode_ar=ar, refine_iters=ri, rounding='bf16')
    ref = g[f'logits.{mode}']
    assert list(got.shape) == list(ref.shape)
    err, msg = report(f'{name} {mode} bf16 logits vs rounding-aware oracle', got, want)
    gap, gmsg = report(f'{name} {mode} bf16 logits vs exact fp32 reference', got, ref)
    assert err <= 3e-2, msg
    assert gap <= 6e-2, gmsg
    # decisions: identical wherever the reference decision is not a near-tie at bf16 resolution
    top2 = ref.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2 * 6e-2
    if ar:      # autoregressive feedback: only the prefix before the first near-tie is comparable position by position
        safe = safe.int().cumprod(-1).bool()
    agree = (got.argmax(-1) == ref.argmax(-1))
    print(f'[{name} {mode} bf16] argmax agreement overall {agree.float().mean().item():.4f}, '
          f'comparable positions {int(safe.sum())}/{safe.numel()}')
    assert bool(agree[safe].all()), 'argmax differs from the fp32 reference at a position with a clear margin'
    strings, _ = m.tokenizer.decode(got.softmax(-1))
    same = sum(a == b for a, b in zip(strings, meta['modes'][mode]['strings']))
    print(f'[{name} {mode} bf16] strings identical to the reference: {same}/{len(strings)}')


def test_batch1_and_batch_invariance(name, models, golden):
    g, _ = golden(name)
    m = models['fp32']
    got1 = _run(m, g['images'][:1].to(DEV), 'ar1')
    err, msg = report(f'{name} batch-1 ar1 fp32 vs reference', got1, g['logits.ar1.batch1'])
    assert err <= 1e-3, msg
    got8 = _run(m, g['images'].to(DEV), 'ar1')
    assert (got8[:1] - got1).abs().max() <= 1e-5        # same image alone or in a batch


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_full_batch_512_replicas(name, models, golden, precision):
    """BASELINE config sizes: 512 crops per GPU.  64 copies of the 8 golden crops must all reproduce the 8-crop result
    (each image is independent; catches tile/row-remap bugs that only appear with many m-tiles)."""
    g, _ = golden(name)
    m = models[precision]
    small = _run(m, g['images'].to(DEV), 'ar1')
    big = _run(m, g['images'].repeat(64, 1, 1, 1).to(DEV), 'ar1')
    assert big.shape == (512, 26, 95)
    d = (big.view(64, 8, 26, 95) - small.unsqueeze(0)).abs().max().item()
    print(f'[{name} {precision} batch-512 replicas] max|d| vs batch-8 run {d:.3e}')
    assert d <= (1e-5 if precision == 'fp32' else 1e-5)   # big and small GEMM tiles accumulate K in the same order


def test_teacher_forced_decode_logits_bf16(name, models, golden):
    """Per-stage hook through the C ABI: one full-context pass with the oracle's own AR tokens (no feedback)."""
    g, _ = golden(name)
    cfg, sd = CONFIGS[name], synth_state_dict(CONFIGS[name], 0)
    images = g['images'][:4]
    with torch.inference_mode():
        tr = O.Trace()
        O.forward(sd, cfg, images, 25, decode_ar=True, refine_iters=0, trace=tr)
        causal = torch.triu(torch.ones(26, 26, dtype=torch.bool), 1)
        for precision, tol in (('fp32', 1e-3), ('bf16', 3e-2)):
            m = models[precision]
            m.model.encode(images.to(DEV))
            got = m.model.decode_logits(tr.ar_tokens, 0, 26, None, causal).cpu()
            mem = O.encode(sd, cfg, images, rounding=None if precision == 'fp32' else 'bf16')
            pos_q = sd['pos_queries'].expand(4, -1, -1)
            want = O.head(sd, O.decode(sd, cfg, tr.ar_tokens, mem, causal, None, pos_q, causal,
                                       rounding=None if precision == 'fp32' else 'bf16'),
                          None if precision == 'fp32' else 'bf16')
            err, msg = report(f'{name} teacher-forced decode {precision}', got, want)
            assert err <= tol, msg


def test_random_crops_fp32_statistics(name, models):
    """Un-selected random crops (no margin filtering): error bar must still hold; argmax agreement is reported and
    must be identical wherever the oracle's top-1/top-2 margin exceeds the tolerance."""
    cfg, sd = CONFIGS[name], synth_state_dict(CONFIGS[name], 0)
    images = synth_images(32, cfg, seed=777)
    m = models['fp32']
    with torch.inference_mode():
        tr = O.Trace()
        want = O.forward(sd, cfg, images, 25, decode_ar=True, refine_iters=0, trace=tr)
    m.model.encode(images.to(DEV))
    causal = torch.triu(torch.ones(26, 26, dtype=torch.bool), 1)
    got = m.model.decode_logits(tr.ar_tokens, 0, 26, None, causal).cpu()
    err, msg = report(f'{name} 32 random crops teacher-forced fp32', got, want)
    assert err <= 1e-3, msg
    top2 = want.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2e-3
    assert torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe])


def test_hub_config1_tiny_nar_batch1(golden):
    """BASELINE config 1: PARSeq-Ti, batch 1, NAR, refine 0 through hubconf — output contract [1, 26, 95]."""
    g, _ = golden('parseq-tiny')
    m = torch.hub.load('.', 'parseq_tiny', source='local', decode_ar=False, refine_iters=0, precision='fp32')
    m.model.load_state_dict(synth_state_dict(CONFIGS['parseq-tiny'], 0))
    m = m.eval().to(DEV)
    with torch.inference_mode():
        out = m(g['images'][:1].to(DEV))
    assert out.shape == (1, 26, 95)
    err, msg = report('config-1 tiny NAR batch-1', out.cpu(), g['logits.nar0'][:1])
    assert err <= 1e-3, msg


def test_test_step_contract(name, models, golden):
    g, meta = golden(name)
    m = models['fp32']
    m.model.decode_ar, m.model.refine_iters = True, 1
    labels = [s.lower() for s in meta['modes']['ar1']['strings']]
    res = m.test_step((g['images'].to(DEV), labels), 0)['output']
    assert res.num_samples == 8 and 0 <= res.correct <= 8 and res.loss is None
    want_conf = sum(meta['modes']['ar1']['confidence'])
    assert abs(res.confidence - want_conf) <= 1e-2 * max(1.0, want_conf)


@pytest.mark.parametrize('precision', ['fp32', 'bf16', 'bf16x3'])
def test_uint8_input_is_normalised_in_the_patch_embed(name, models, precision):
    """Row N2: raw uint8 pixels through parseq_forward(images_dtype=PARSEQ_U8) == the reference transform's
    ToTensor + Normalize(0.5, 0.5) (oracle.normalize_u8) fed as a float tensor, bit for bit (every pixel value occurs)."""
    m = models[precision]
    m.model.decode_ar, m.model.refine_iters = True, 1
    g = torch.Generator().manual_seed(99)
    u8 = torch.randint(0, 256, (9, 3, 32, 128), generator=g, dtype=torch.uint8)
    u8[0].view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
    ref_in = O.normalize_u8(u8)
    if precision == 'bf16':
        ref_in = ref_in.bfloat16()
    with torch.inference_mode():
        got = m(u8.to(DEV), 25).float().cpu()
        want = m(ref_in.to(DEV), 25).float().cpu()
    assert torch.equal(got, want)
    with torch.inference_mode():
        mem_u8 = m.model.encode(u8.to(DEV)).cpu()
        mem_f = m.model.encode(ref_in.to(DEV)).cpu()
    assert torch.equal(mem_u8, mem_f)


def test_patch_head_of_the_bf16x3_encoder(golden, monkeypatch):
    """encoder_blocks_x3.h patch_head_x3 (the (4, 8) patch embedding as the head of the bf16x3 one-launch encoder: pixels split into
    bf16 pairs, weight stages from the hi | lo pack, accumulators started from the pos_embed + bias table) against the same model with
    the patch embedding as its own GEMM launch (PARSEQ_NO_FUSED_HEAD=1): same operands and products, only the fp32 summation order
    differs (bias + pos_embed first instead of last) — and both within the encoder bar of the reference's own memory."""
    g, _ = golden('parseq')
    images = g['images'].to(DEV)
    fused = make_model('parseq', 'bf16x3')
    with torch.inference_mode():
        a = fused.model.encode(images).cpu()
        a16 = fused.model.encode(images.bfloat16()).cpu()      # bf16 pixels: exact in f32, lo planes zero
    monkeypatch.setenv('PARSEQ_NO_FUSED_HEAD', '1')
    sep = make_model('parseq', 'bf16x3')                         # the switch is read when a plan is created
    with torch.inference_mode():
        b = sep.model.encode(images).cpu()
        b16 = sep.model.encode(images.bfloat16()).cpu()
    for tag, x, y in (('f32 pixels', a, b), ('bf16 pixels', a16, b16)):
        d, msg = report(f'bf16x3 patch head in-launch vs separate GEMM ({tag})', x, y)
        assert 0 < d <= 2e-4, msg                               # measured 4.9e-5 on values of O(4): twelve blocks downstream of a 1e-7 difference in x
    err, msg = report('bf16x3 memory with the in-launch head vs reference', a, g['memory'])
    assert err <= 5e-4, msg


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_config3_batch_1024_ar_two_refinements(golden, precision):
    """BASELINE configs[3]: PARSeq-S, 94-char set, max_label_length 25, AR + 2 refinement iterations, batch 1024 on one GPU.
    128 copies of the 8 golden crops: fp32 mode against the reference's own logits (bar 1e-3), bf16 against its batch-8 run."""
    g, _ = golden('parseq')
    m = make_model('parseq', precision)
    big = _run(m, g['images'].repeat(128, 1, 1, 1).to(DEV), 'ar2')
    assert big.shape == (1024, 26, 95)
    want = g['logits.ar2'] if precision == 'fp32' else _run(m, g['images'].to(DEV), 'ar2')
    d = (big.view(128, 8, 26, 95) - want.unsqueeze(0)).abs().max().item()
    print(f'[config3 {precision}] batch 1024 AR+2: max|d| {d:.3e}')
    assert d <= (1e-3 if precision == 'fp32' else 1e-5)
    if precision == 'fp32':
        assert torch.equal(big.view(128, 8, 26, 95).argmax(-1), want.argmax(-1).unsqueeze(0).expand(128, -1, -1))


@pytest.mark.parametrize('batch', [1, 7, 17, 100, 333])
def test_ragged_batch_sizes_bf16(name, models, golden, batch):
    """Row tails of every kernel (16-row step tiles, 64 / 128-row GEMM tiles, 4-images-per-block row ops): image i of a
    ragged batch must equal the same crop's result in the 8-crop batch."""
    g, _ = golden(name)
    m = models['bf16']
    idx = torch.arange(batch) % 8
    small = _run(m, g['images'].to(DEV), 'ar1')
    got = _run(m, g['images'][idx].to(DEV), 'ar1')
    d = (got - small[idx]).abs().max().item()
    print(f'[{name} ragged batch {batch}] max|d| {d:.3e}')
    assert d <= 1e-5


def test_weight_update_refreshes_native_state(golden):
    """load_state_dict / in-place edits after the first forward must reach the device twin (packed weights, decoder tables,
    fragment-packed step weights): change the weights, compare with the oracle on the NEW weights."""
    g, _ = golden('parseq')
    for precision, tol in (('fp32', 1e-3), ('bf16', 6e-2)):
        m = make_model('parseq', precision)
        images = g['images'].to(DEV)
        before = _run(m, images, 'ar1')
        sd2 = synth_state_dict(CONFIGS['parseq'], 5)                     # a different weight set
        m.model.load_state_dict(sd2)
        after = _run(m, images, 'ar1')
        with torch.inference_mode():
            want = O.forward(sd2, CONFIGS['parseq'], g['images'], None, decode_ar=True, refine_iters=1)
        assert (before - after).abs().max() > 0.1                        # the update is visible ...
        d, msg = report(f'after load_state_dict {precision}', after, want)
        assert d <= tol, msg                                             # ... and complete
        with torch.no_grad():
            m.model.head.bias.add_(1.0)                                  # in-place edit of one tensor
        shifted = _run(m, images, 'nar0')
        with torch.no_grad():
            m.model.head.bias.sub_(1.0)          # (edits through `.data` do not bump the tensor version and are not tracked)
        base = _run(m, images, 'nar0')
        assert torch.allclose(shifted - base, torch.ones_like(base), atol=2e-2 if precision == 'bf16' else 1e-5)
        # a write through `.data` (EMA / checkpoint-averaging idiom) is invisible to the version counters: mark_dirty() is the
        # explicit way to have it uploaded
        m.model.head.bias.data += 1.0
        stale = _run(m, images, 'nar0')
        assert torch.equal(stale, base)
        m.model.mark_dirty()
        seen = _run(m, images, 'nar0')
        m.model.head.bias.data -= 1.0
        m.model.mark_dirty()
        assert torch.allclose(seen - base, torch.ones_like(base), atol=2e-2 if precision == 'bf16' else 1e-5)
        assert torch.equal(_run(m, images, 'nar0'), base)


def test_encoder_tail_in_launch_vs_separate(golden, monkeypatch):
    """parseq_forward's encoder tail (final LayerNorm + decoder K / V projection of memory inside the one-launch encoder,
    encoder_blocks.h kv_phase) against the same model with the tail as its own launches (PARSEQ_NO_FUSED_TAIL=1): the two round to
    bf16 at the same points and differ only in fp32 accumulation order."""
    g, _ = golden('parseq')
    images = g['images'].to(DEV).repeat(4, 1, 1, 1)
    fused = make_model('parseq', 'bf16')
    a = _run(fused, images, 'nar0')
    monkeypatch.setenv('PARSEQ_NO_FUSED_TAIL', '1')
    sep = make_model('parseq', 'bf16')                       # the switch is read when a plan is created
    b = _run(sep, images, 'nar0')
    d, msg = report('encoder tail in-launch vs separate launches (bf16, NAR)', a, b)
    assert d <= 2e-2, msg
    assert (a.argmax(-1) == b.argmax(-1)).float().mean() >= 0.99


def test_bf16x3_forward_eight_waves_equals_four_waves_bit_for_bit(golden, monkeypatch):
    """parseq_forward in the exact-tolerance mode launches the encoder on eight waves of 16 rows per workgroup (encoder_blocks_x3w.h, round 5); PARSEQ_X3_FOUR_WAVES=1 puts
    round 4's four-wave kernel back.  The two accumulate the same products in the same order, so the whole forward — patch head from the raw crops, twelve blocks, 24-bit K / V
    tail, AR loop, refinement — must agree bit for bit: logits, and for u8 crops as well."""
    g, _ = golden('parseq')
    images = g['images'].to(DEV).repeat(9, 1, 1, 1)[:20]
    w8 = make_model('parseq', 'bf16x3')
    a = _run(w8, images, 'ar1')
    u8 = ((images * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8)
    with torch.inference_mode():
        a8 = w8(u8, 25).float().clone()
    monkeypatch.setenv('PARSEQ_X3_FOUR_WAVES', '1')
    w4 = make_model('parseq', 'bf16x3')                      # the switch is read when a plan is created
    b = _run(w4, images, 'ar1')
    with torch.inference_mode():
        b8 = w4(u8, 25).float().clone()
    assert torch.equal(a, b), report('bf16x3 forward: eight waves vs four waves', a, b)[1]
    assert torch.equal(a8, b8), report('bf16x3 forward on u8 crops: eight waves vs four waves', a8, b8)[1]


def test_kv_rows_24_bit_vs_f32(golden, monkeypatch):
    """bf16x3 mode, PARSeq-S: the one-launch encoder's tail leaves the decoder's K / V rows as 24-bit floats (16 significant bits in a
    u16 + a u8 plane: decoder_attn.h F24; 25 % less of the AR loop's HBM stream) against the same model with f32 rows
    (PARSEQ_NO_KV24=1) — AR loop (dec_cross_attn_ar24_kernel) and refinement pass (dec_cross_attn_multi_mfma_x3_kernel<true>).  The CPU
    study (profiles/r04_cheap_exact_study.md) puts the format's cost at 1.6e-5 on the logits; both must stay within 1e-3 of the
    reference's own outputs.  encode() + decode() on the same model keeps f32 rows (generic GEMM) and must still agree."""
    g, _ = golden('parseq')
    images = g['images'].to(DEV)
    m24 = make_model('parseq', 'bf16x3')
    a_nar, a_ar = _run(m24, images, 'nar1'), _run(m24, images, 'ar1')
    for mode, got in (('nar1', a_nar), ('ar1', a_ar)):
        d, msg = report(f'24-bit K / V rows vs reference golden ({mode})', got, g[f'logits.{mode}'])
        assert d <= 1e-3, msg
        assert torch.equal(got.argmax(-1).cpu(), g[f'logits.{mode}'].argmax(-1)), msg
    monkeypatch.setenv('PARSEQ_NO_KV24', '1')
    m32 = make_model('parseq', 'bf16x3')                     # the switch is read when a plan is created
    b_nar, b_ar = _run(m32, images, 'nar1'), _run(m32, images, 'ar1')
    d, msg = report('24-bit vs f32 K / V rows (bf16x3, NAR + 1 refinement)', a_nar, b_nar)
    assert 0 < d <= 1e-4, msg                                # different storage (d > 0: the 24-bit path really ran), far inside the tolerance
    d, msg = report('24-bit vs f32 K / V rows (bf16x3, AR + 1 refinement)', a_ar, b_ar)
    assert d <= 1e-4, msg
    # the reference idiom on the 24-bit model: encode() leaves f32 rows behind (generic GEMM), decode() must read THOSE
    with torch.inference_mode():
        mem = m24.model.encode(images)
        tgt = torch.full((images.shape[0], 1), m24.tokenizer.bos_id, dtype=torch.long, device=DEV)
        out = m24.model.head(m24.model.decode(tgt, mem, tgt_query=m24.model.pos_queries[:, :1]))      # AR step 0: context <bos>, query 0
        again = _run(m24, images, 'ar1')                     # and a forward afterwards switches back to 24-bit rows
    d, msg = report('decode() after encode() on a model whose forward uses 24-bit rows', out[:, 0].float().cpu(), g['logits.ar0'][:, 0])
    assert d <= 1e-3, msg
    assert torch.equal(again, a_ar)


@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
def test_ar_step_split_over_workgroups_vs_one_workgroup(golden, models, monkeypatch, name, precision):
    """The AR step's out_proj -> norm1 -> q-projection chain split over DS_QS workgroups per row tile (decoder_step.h
    dec_step_mid_kernel<.., QS = 3>: column slices of out_proj, the q-projection as a K-split whose partial sums the cross-attention
    kernel adds up, LayerNorm finished behind the product from a mean known in advance and exchanged column sums) — what a forward that has
    the device to itself runs (PARSEQ_FLAG_LATENCY: the default call, no `slot`) — against the same chain on one workgroup per tile (an
    explicit slot: batches in flight).  Same products, different summation order and a LayerNorm scaled after instead of before the
    product: the bf16x3 logits may differ by rounding only; both must stay within the tolerance of the reference's own outputs.
    PARSEQ_NO_QSPLIT=1 (read when a plan is created) turns the split off altogether: then the two calls are the same computation."""
    g, _ = golden(name)
    images = g['images'].to(DEV)
    m = models[precision]

    def narrow(mode):
        ar, ri, ml = MODES[mode]
        m.model.decode_ar, m.model.refine_iters = ar, ri
        with torch.inference_mode():
            out = m(images, ml, slot=0)
        torch.cuda.synchronize()
        return out.float().cpu()
    a0, a1 = _run(m, images, 'ar0'), _run(m, images, 'ar1')
    b0, b1 = narrow('ar0'), narrow('ar1')
    if precision == 'bf16x3':
        for mode, got in (('ar0', a0), ('ar1', a1), ('ar0', b0), ('ar1', b1)):
            d, msg = report(f'AR step vs reference golden ({name}, {mode})', got, g[f'logits.{mode}'])
            assert d <= 1e-3, msg
            assert torch.equal(got.argmax(-1).cpu(), g[f'logits.{mode}'].argmax(-1)), msg
        d, msg = report(f'split vs one-workgroup AR step ({name}, bf16x3, AR)', a0, b0)
        assert 0 < d <= 1e-4, msg                            # d > 0: the split path really ran
        d, msg = report(f'split vs one-workgroup AR step ({name}, bf16x3, AR + 1 refinement)', a1, b1)
        assert d <= 1e-4, msg
    else:
        # bf16 operands: the two forms round different intermediates ((x - m) ln_w against the normalised row); both sit at the mode's
        # own distance from the exact result, so they are compared through it
        ref = g['logits.ar0']
        da, _ = report(f'split AR step vs golden ({name}, bf16)', a0, ref)
        db, _ = report(f'one-workgroup AR step vs golden ({name}, bf16)', b0, ref)
        assert torch.isfinite(a0).all() and not torch.equal(a0, b0)
        assert da <= max(2.0 * db, 0.05), (da, db)
    if precision == 'bf16x3':
        # ragged row tiles: batches that are not a multiple of the 16 rows a workgroup owns (the column sums and partials of the rows past
        # the batch must be neither written nor read)
        big = images.repeat(5, 1, 1, 1)
        m.model.decode_ar, m.model.refine_iters = True, 0
        with torch.inference_mode():
            ref25 = m(images, 25).float().cpu()
        for B in (1, 5, 17, 33):
            x = big[:B].contiguous()
            with torch.inference_mode():
                wide, narrow_ = m(x, 25).float().cpu(), m(x, 25, slot=0).float().cpu()
            d, msg = report(f'split vs one-workgroup AR step, batch {B} ({name})', wide, narrow_)
            assert d <= 1e-4 and torch.equal(wide[:min(B, 8)], ref25[:min(B, 8)]), msg     # and the same rows as in the batch of 8
    monkeypatch.setenv('PARSEQ_NO_QSPLIT', '1')
    off = make_model(name, precision)
    assert torch.equal(_run(off, images, 'ar0'), b0)


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
def test_slots_and_streams_give_identical_results(models, golden, name, precision):
    """`slot=k` workspaces on separate streams (bench.py --streams 2) must not interfere: two batches in flight reproduce the
    one-at-a-time results bit for bit (both timed modes of bench.py)."""
    g, _ = golden(name)
    m = models[precision]
    m.model.decode_ar, m.model.refine_iters = True, 1
    a = g['images'].repeat(8, 1, 1, 1).to(DEV)
    b = a.flip(0).contiguous()
    with torch.inference_mode():
        ref_a, ref_b = m(a, 25, slot=0).clone(), m(b, 25, slot=0).clone()      # an explicit slot: the in-flight form of the AR step
        torch.cuda.synchronize()          # the reference runs used slot 0 on the default stream: drain before reusing it elsewhere
        s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for it in range(3):
            with torch.cuda.stream(s0):
                oa = m(a, 25, slot=0)
            with torch.cuda.stream(s1):
                ob = m(b, 25, slot=1)
            outs.append((oa, ob))
        torch.cuda.synchronize()
    for oa, ob in outs:
        assert torch.equal(oa, ref_a) and torch.equal(ob, ref_b)


def test_validation_step_loss(golden):
    """base.py:112-143 with validation=True: forward_logits_loss (base.py:194-201) = logits for max_len = longest label, mean
    cross-entropy over non-<pad> targets (device kernel), loss_numel; checked against the oracle's logits + F.cross_entropy."""
    g, meta = golden('parseq')
    m = make_model('parseq', 'fp32')
    labels = ['hello', 'MI355X', 'a', 'parseq-amd', 'x' * 25, '0123', 'Zz', '!?']
    images = g['images'].to(DEV)
    logits, loss, numel = m.forward_logits_loss(images, labels)
    torch.cuda.synchronize()
    targets = m.tokenizer.encode(labels)[:, 1:]
    assert logits.shape == (8, targets.shape[1], 95) and int(numel) == sum(len(s) + 1 for s in labels)
    with torch.inference_mode():
        want_logits = O.forward(synth_state_dict(CONFIGS['parseq'], 0), CONFIGS['parseq'], g['images'], targets.shape[1] - 1,
                                decode_ar=True, refine_iters=1)
    want_loss, want_numel = O.validation_loss(want_logits, targets, m.pad_id)
    assert (logits.cpu() - want_logits).abs().max() <= 1e-3
    assert abs(float(loss) - float(want_loss)) <= 1e-4 * max(1.0, float(want_loss)) and int(want_numel) == int(numel)
    # the kernel alone, bit-level semantics: ignored rows, every row ignored -> NaN like torch
    lg = torch.randn(40, 95) * 3
    tg = torch.randint(0, 95, (40,))
    tg[::3] = 96
    from parseq_amd import _native
    out_l, out_n = torch.empty((), device=DEV), torch.empty((), dtype=torch.int32, device=DEV)
    ws = torch.empty(40, device=DEV)
    lgd, tgd = lg.to(DEV), tg.to(torch.int32).to(DEV)
    _native.check(_native.lib().parseq_cross_entropy(_native.ptr(lgd), _native.ptr(tgd), 40, 95, 96, _native.ptr(out_l), _native.ptr(out_n),
                                                     _native.ptr(ws), _native.stream_ptr()))
    ref = torch.nn.functional.cross_entropy(lg, tg, ignore_index=96)
    assert abs(float(out_l) - float(ref)) <= 2e-6 * float(ref) + 1e-6 and int(out_n) == int((tg != 96).sum())
    res = m.validation_step((images, labels), 0)['output']
    assert res.num_samples == 8 and abs(float(res.loss) - float(want_loss)) <= 1e-4 * max(1.0, float(want_loss)) and int(res.loss_numel) == int(numel)


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
def test_batch_520_tail_tiles_bf16(golden, precision):
    """520 crops = 520 workgroups of the one-launch encoder on 256 CUs (two whole rounds and eight left over): an image's result
    must not depend on how many other images are in the batch or where it sits — bit for bit the batch-8 result.  (Without
    the one-launch encoder, PARSEQ_NO_FUSED_BLOCKS, the tail tiles of the fused MLP kernel go through the per-op kernels and
    agree within the bf16 bar only.)"""
    g, _ = golden('parseq')
    m = make_model('parseq', precision)
    idx = torch.arange(520) % 8
    small = _run(m, g['images'].to(DEV), 'ar1')
    got = _run(m, g['images'][idx].to(DEV), 'ar1')
    d = (got - small[idx]).abs().amax(dim=(1, 2))
    # bf16: bit for bit up to the accumulation order of the refinement pass's GEMMs.  bf16x3: an image's row is the exact result to
    # ~1e-5 whatever the batch; the refinement GEMMs pick their tile shape from M (208 rows here, 13 520 there), so the two runs
    # differ by fp32 accumulation order (measured 1.3e-5)
    assert d[:512].max().item() <= (1e-5 if precision == 'bf16' else 1e-4)
    assert d[512:].max().item() <= (6e-2 if precision == 'bf16' else 1e-4)


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('model_name,batch', [('parseq', 512), ('parseq', 333), ('parseq-tiny', 512), ('vitstr', 512), ('parseq-patch16-224', 40)])
def test_repeated_runs_are_bit_identical(model_name, batch, precision):
    """Every kernel synchronises its LDS rings with counted `s_waitcnt vmcnt` and barriers it places itself; a missing wait
    would show up as run-to-run differences long before it shows up as a visible error.  Eight runs, same input, in both matrix-core
    modes (bf16x3: the one-launch encoder of encoder_blocks_x3.h, the fused AR step on bf16 pairs, and — parseq-tiny, vitstr, 333
    crops — the pre-split PAIRS GEMMs at the shapes where two workgroups share a CU)."""
    if model_name == 'vitstr':
        from oracle import vitstr_oracle as V
        from parseq_amd import create_model
        m = create_model('vitstr', precision=precision)
        m.model.load_state_dict(V.synth_state_dict(V.vitstr_config(), 0))
        m = m.eval().to(DEV)
        size = (32, 128)
    else:
        m = make_model(model_name, precision)
        size = tuple(CONFIGS[model_name].img_size)
    x = (torch.rand(batch, 3, *size, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(DEV)
    with torch.inference_mode():
        first = m(x, 25).clone()
        for _ in range(7):
            assert torch.equal(m(x, 25), first)


def test_decode_and_head_reference_idiom(name, models, golden):
    """`model.head(model.decode(tgt, memory, ...))` — the reference's own call pattern (model.py:138,152,167; system.py:149-150):
    decoder output after decoder.norm and a callable head, against the oracle; single-position and sliced-query forms too."""
    g, _ = golden(name)
    cfg, sd = CONFIGS[name], synth_state_dict(CONFIGS[name], 0)
    images = g['images'][:4]
    m = models['fp32']
    with torch.inference_mode():
        tr = O.Trace()
        O.forward(sd, cfg, images, 25, decode_ar=True, refine_iters=0, trace=tr)
        mem_o = O.encode(sd, cfg, images)
        causal = torch.triu(torch.ones(26, 26, dtype=torch.bool), 1)
        pos_q = sd['pos_queries'].expand(4, -1, -1)
        want_h = O.decode(sd, cfg, tr.ar_tokens, mem_o, causal, None, pos_q, causal)
        memory = m.model.encode(images.to(DEV))
        tgt = tr.ar_tokens.to(DEV)
        hid = m.model.decode(tgt, memory, tgt_mask=causal.to(DEV), tgt_query_mask=causal.to(DEV))
        logits = m.model.head(hid)
        assert hid.shape == (4, 26, cfg.embed_dim)
        assert (hid.cpu() - want_h).abs().max() <= 1e-3
        assert (logits.cpu() - O.head(sd, want_h)).abs().max() <= 1e-3
        assert (logits - m.model.decode_logits(tr.ar_tokens, 0, 26, None, causal)).abs().max() <= 1e-4
        # AR-step form: context tgt[:, :j], one query pos_queries[:, i:j] with its mask row (model.py:126-137)
        i, j = 5, 6
        hid1 = m.model.decode(tgt[:, :j], memory, tgt_query=m.model.pos_queries[:, i:j], tgt_query_mask=causal[i:j, :j].to(DEV))
        assert hid1.shape == (4, 1, cfg.embed_dim) and (hid1[:, 0] - hid[:, i]).abs().max() <= 1e-4
        # arbitrary tgt_query tensors (model.py:100-102): a per-image query stream that is NOT a slice of pos_queries
        gq = torch.Generator().manual_seed(5)
        uq = 0.5 * torch.randn(4, 7, cfg.embed_dim, generator=gq)
        qmask = torch.rand(7, 26, generator=gq) < 0.3
        qmask[:, 0] = False                                    # keep <bos> visible: no fully masked row
        want_u = O.decode(sd, cfg, tr.ar_tokens, mem_o, causal, None, uq, qmask)
        hid_u = m.model.decode(tgt, memory, tgt_query=uq.to(DEV), tgt_query_mask=qmask.to(DEV))
        assert hid_u.shape == (4, 7, cfg.embed_dim)
        assert (hid_u.cpu() - want_u).abs().max() <= 1e-3
        # the same values handed over as a copy of pos_queries must equal the table-served result
        hid_c = m.model.decode(tgt, memory, tgt_query=m.model.pos_queries.detach().clone().expand(4, -1, -1), tgt_query_mask=causal.to(DEV))
        assert (hid_c - hid).abs().max() <= 1e-4
        # a caller-supplied `memory` is honoured (model.py:89), not replaced by the last encode's: decode against the memory
        # of OTHER images after an intervening encode, then against an edited copy
        other = g['images'][4:8]
        mem_other_o = O.encode(sd, cfg, other)
        mem_other = m.model.encode(other.to(DEV))             # the plan now caches K / V of `other`
        hid_back = m.model.decode(tgt, memory, tgt_query_mask=causal.to(DEV))      # ... but `memory` is what was asked for
        assert (hid_back - hid).abs().max() <= 1e-5
        want_o = O.decode(sd, cfg, tr.ar_tokens, mem_other_o, causal, None, pos_q, causal)
        hid_o = m.model.decode(tgt, mem_other, tgt_query_mask=causal.to(DEV))
        assert (hid_o.cpu() - want_o).abs().max() <= 1e-3
        edited = mem_other.clone()
        edited[:, :64] = 0
        mem_e = mem_other_o.clone()
        mem_e[:, :64] = 0
        want_e = O.decode(sd, cfg, tr.ar_tokens, mem_e, causal, None, pos_q, causal)
        assert (m.model.decode(tgt, edited, tgt_query_mask=causal.to(DEV)).cpu() - want_e).abs().max() <= 1e-3
        mem_other.mul_(0.5)                                    # in-place edit of the tensor encode() returned: version bump -> re-projected
        want_h2 = O.decode(sd, cfg, tr.ar_tokens, 0.5 * mem_other_o, causal, None, pos_q, causal)
        assert (m.model.decode(tgt, mem_other, tgt_query_mask=causal.to(DEV)).cpu() - want_h2).abs().max() <= 1e-3
        # out-of-range token ids are clamped on the device, never used as raw table indices
        bad = tgt.clone()
        bad[0, 3] = 10 ** 6
        bad[1, 2] = -5
        assert torch.isfinite(m.model.decode(bad, memory, tgt_query_mask=causal.to(DEV))).all()


# ---- BASELINE.json configurations on DISTINCT data (configs[1]: 512 crops AR+1; configs[3]: 1024 crops AR+2) -----------------
# 512 / 1024 different seeded crops.  Two checks per exact-tolerance mode (fp32, bf16x3), both against the CPU oracle on every crop:
#   (1) arithmetic: every decoder pass the reference runs — the AR pass and each refinement — is re-run teacher-forced with the
#       ORACLE's own context tokens (parseq_decode_logits), so no decision feeds back: |dlogit| <= 1e-3 on all positions of all crops;
#   (2) end to end: model(images) — the decisions now feed back, and a decision may legitimately differ where the oracle's own
#       top-1 / top-2 margin is inside the arithmetic tolerance (random-init weights: median margin 0.3, some below 1e-3 among
#       13 312 positions).  Every crop whose final logits deviate by more than 1e-3 must have such a near-tie in the oracle's
#       trace; every other crop must meet 1e-3 and argmax identity; the fraction of identical strings is asserted.
# The bf16 mode is held to its stated bars against the rounding-aware oracle, string agreement asserted against a floor.
def _oracle_traced(sd, cfg, images, refine_iters, rounding=None, chunk=64):
    outs, traces = [], []
    with torch.inference_mode():
        for i in range(0, images.shape[0], chunk):
            tr = O.Trace()
            outs.append(O.forward(sd, cfg, images[i:i + chunk], 25, decode_ar=True, refine_iters=refine_iters, rounding=rounding, trace=tr))
            traces.append(tr)
    cat = lambda f: torch.cat([f(t) for t in traces])      # noqa: E731
    tr = O.Trace(ar_logits=cat(lambda t: t.ar_logits), ar_tokens=cat(lambda t: t.ar_tokens),
                 refine_logits=[cat(lambda t, k=k: t.refine_logits[k]) for k in range(refine_iters)],
                 refine_tokens=[cat(lambda t, k=k: t.refine_tokens[k]) for k in range(refine_iters)])
    return torch.cat(outs), tr


def _margin(logits):
    top2 = logits.topk(2, -1).values
    return top2[..., 0] - top2[..., 1]


@pytest.mark.parametrize('batch,refine', [(512, 1), (1024, 2)])
def test_baseline_configs_distinct_crops(batch, refine):
    name = 'parseq'
    cfg, sd = CONFIGS[name], synth_state_dict(CONFIGS[name], 0)
    images = synth_images(batch, cfg, seed=20250924 + batch)
    assert images.flatten(1).unique(dim=0).shape[0] == batch              # really distinct crops
    torch.set_num_threads(min(32, torch.get_num_threads()))
    want, tr = _oracle_traced(sd, cfg, images, refine)
    causal = torch.triu(torch.ones(26, 26, dtype=torch.bool), 1)
    cloze = causal.clone()
    cloze[torch.triu(torch.ones(26, 26, dtype=torch.bool), 2)] = False
    # a crop is "decided" when no pass of the oracle has a position whose two best classes are closer than twice the tolerance
    tight = (_margin(tr.ar_logits) < 2e-3).any(-1)
    for k in range(refine - 1):                       # the last refinement's decisions feed nothing
        tight |= (_margin(tr.refine_logits[k]) < 2e-3).any(-1)
    print(f'[batch {batch}] crops with a near-tie (< 2e-3) somewhere in the oracle trace: {int(tight.sum())}')
    for precision in ('fp32', 'bf16x3'):
        m = make_model(name, precision, refine_iters=refine)
        with torch.inference_mode():
            x = images.to(DEV)
            got = m(x, 25).float().cpu()
            m.model.encode(x)                                                 # (1) teacher-forced passes
            ar = m.model.decode_logits(tr.ar_tokens, 0, 26, None, causal).cpu()
            err_ar, msg = report(f'{precision} batch {batch} teacher-forced AR pass, distinct crops vs CPU oracle', ar, tr.ar_logits)
            assert err_ar <= 1e-3, msg
            for k in range(refine):
                toks = tr.refine_tokens[k]
                kpm = (toks == cfg.eos_id).int().cumsum(-1) > 0
                rf = m.model.decode_logits(toks, 0, 26, kpm, cloze).cpu()
                err_rf, msg = report(f'{precision} batch {batch} teacher-forced refinement {k}, distinct crops vs CPU oracle', rf, tr.refine_logits[k])
                assert err_rf <= 1e-3, msg
        assert got.shape == want.shape == (batch, 26, 95)                     # (2) end to end
        per_crop = (got - want).abs().flatten(1).max(1).values
        off = per_crop > 1e-3
        print(f'[{precision} batch {batch}] end to end: max|d| over decided crops {per_crop[~tight].max():.3e}, '
              f'crops beyond 1e-3: {int(off.sum())} (all must be near-tie crops)')
        assert not bool((off & ~tight).any()), f'{precision}: a crop without any near-tie deviates by {per_crop[off & ~tight].max():.3e}'
        clear = (_margin(want) > 2e-3) & ~tight[:, None]                       # positions whose own decision is not a near-tie either
        assert bool((got.argmax(-1) == want.argmax(-1))[clear].all())
        s_got, _ = m.tokenizer.decode(got.softmax(-1))
        s_want, _ = m.tokenizer.decode(want.softmax(-1))
        same = sum(a == b for a, b in zip(s_got, s_want)) / batch
        print(f'[{precision} batch {batch}] strings identical to the oracle: {same:.4f}')
        assert same >= 0.99, f'{precision}: only {same:.4f} of the strings equal the oracle'
        del m
    # bf16: same rounding points as the rounding-aware oracle
    want16, _ = _oracle_traced(sd, cfg, images, refine, rounding='bf16')
    m = make_model(name, 'bf16', refine_iters=refine)
    with torch.inference_mode():
        got = m(images.to(DEV), 25).float().cpu()
    err, msg = report(f'{name} bf16 batch {batch} AR+{refine} distinct crops vs rounding-aware oracle', got, want16)
    gap, gmsg = report(f'{name} bf16 batch {batch} AR+{refine} distinct crops vs exact fp32 oracle', got, want)
    # bf16 decisions flip at near-ties: the bars apply to the crops whose strings agree with the respective oracle
    s_got, _ = m.tokenizer.decode(got.softmax(-1))
    s_w16, _ = m.tokenizer.decode(want16.softmax(-1))
    s_want, _ = m.tokenizer.decode(want.softmax(-1))
    agree16 = torch.tensor([a == b for a, b in zip(s_got, s_w16)])
    agree32 = torch.tensor([a == b for a, b in zip(s_got, s_want)])
    e16 = (got - want16).abs().flatten(1).max(1).values
    e32 = (got - want).abs().flatten(1).max(1).values
    print(f'[bf16 batch {batch}] strings identical: {agree16.float().mean():.4f} (rounding-aware oracle), {agree32.float().mean():.4f} (fp32 oracle); '
          f'max|d| on agreeing crops {e16[agree16].max():.3e} / {e32[agree32].max():.3e}')
    assert agree32.float().mean() >= 0.90, 'bf16: fewer than 90 % of the strings equal the fp32 oracle (0.957 measured on 4096 crops)'
    assert agree16.float().mean() >= 0.90
    assert e16[agree16].median() <= 3e-2 and e32[agree32].median() <= 6e-2


@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
def test_early_exit_length_is_the_batch_maximum(golden, precision):
    """With the natural early exit (AR, no refinement, max_length=None) the returned length is the step at which EVERY row of the
    batch holds an EOS (model.py:144-145) — counted on the device by the fused AR step, read back once.  A batch whose images exit at
    different steps must return the maximum of their solo lengths, and every image the values of its solo run over its own length."""
    g, _ = golden('parseq')
    m = make_model('parseq', precision, decode_ar=True, refine_iters=0)
    imgs = g['images'].to(DEV)
    with torch.inference_mode():
        solo = [m(imgs[i:i + 1]).float().cpu() for i in range(8)]
    lens = [int(s.shape[1]) for s in solo]
    order = sorted(range(8), key=lambda i: lens[i])
    assert lens[order[0]] < lens[order[-1]], f'golden crops all exit at the same step {lens}: the test would be vacuous'
    pick = [order[0], order[2], order[5], order[-1]]
    idx = torch.tensor([p for p in pick for _ in range(64)])      # 256 images = 16 row tiles of the step kernels
    with torch.inference_mode():
        got = m(imgs[idx]).float().cpu()
    assert got.shape[1] == max(lens[p] for p in pick), (got.shape, [lens[p] for p in pick])
    for k, p in enumerate(pick):
        L = lens[p]
        assert torch.equal(got[64 * k:64 * k + 64, :L], solo[p].expand(64, -1, -1)), f'group {k} (crop {p}) differs from its solo run'


def test_bf16x3_fused_ar_step_vs_per_op_kernels(monkeypatch):
    """bf16x3 AR loop: the fused step (decoder_step.h on bf16 pairs: mid / cross-attention / mlp launches) against the same mode
    through the per-op kernels (PARSEQ_NO_FUSED_STEP=1), 512 distinct random crops, AR with all 26 steps + 1 refinement.  Both carry
    ~16 mantissa bits per operand, so rows whose greedy decisions agree must agree to ~1e-4; a row may legitimately diverge only where
    a decision is a near-tie at that resolution."""
    x = (torch.rand(512, 3, 32, 128, generator=torch.Generator().manual_seed(17)) * 2 - 1).to(DEV)
    fused = make_model('parseq', 'bf16x3')
    a = _run(fused, x, 'ar1')
    a0 = _run(fused, x, 'ar0_full')
    monkeypatch.setenv('PARSEQ_NO_FUSED_STEP', '1')
    perop = make_model('parseq', 'bf16x3')                      # the switch is read when a plan is created
    b = _run(perop, x, 'ar1')
    b0 = _run(perop, x, 'ar0_full')
    for tag, u, v in (('AR+1', a, b), ('AR+0', a0, b0)):
        same = (u.argmax(-1) == v.argmax(-1)).all(-1)
        d = (u - v).abs().amax(dim=(1, 2))
        print(f'[bf16x3 fused vs per-op {tag}] rows with identical decisions {int(same.sum())}/512, max|d| on them {d[same].max().item():.3e}, '
              f'overall {d.max().item():.3e}')
        assert same.float().mean() >= 0.99
        assert d[same].max().item() <= 2e-4


def test_memory_cache_follows_the_plan_and_the_weights(golden):
    """`decode(tgt, memory)` skips the K / V projection only when THIS plan still holds the projection of THIS tensor made with THESE
    weights (model.py _bind_memory).  Three ways the cached key used to go stale: mark_dirty() releases every plan; switching
    `precision` between encode and decode selects another plan (which may hold another batch's K / V); a weight update re-packs the
    plans under the same tensor.  Each must re-project — checked against decode() of a CLONE of the memory (always projected afresh)."""
    g, _ = golden('parseq-tiny')
    m = make_model('parseq-tiny', 'fp32')
    a, b = g['images'][:4].to(DEV), g['images'][4:8].to(DEV)
    tgt = torch.full((4, 1), m.tokenizer.bos_id, dtype=torch.long, device=DEV)
    with torch.no_grad():                                       # (not inference_mode: inference tensors carry no version and are never cached)
        mem = m.model.encode(a)
        h0 = m.model.decode(tgt, mem).clone()
        assert torch.equal(m.model.decode(tgt, mem), h0)         # cached path
        # (1) mark_dirty(): plans are rebuilt, the projection must be made again (used to fail: 'batch does not match the last encode')
        m.model.mark_dirty()
        assert torch.equal(m.model.decode(tgt, mem), h0)
        # (2) another precision = another plan; let that plan hold ANOTHER batch's K / V of the same size first
        m.model.precision = 'bf16x3'
        m.model.encode(b)
        m.model.precision = 'fp32'
        mem2 = m.model.encode(a)                                 # fp32 plan caches `a`
        m.model.precision = 'bf16x3'
        hx = m.model.decode(tgt, mem2)                           # bf16x3 plan holds `b`: must re-project `a`
        assert (hx - h0).abs().max() <= 1e-3
        assert torch.equal(hx, m.model.decode(tgt, mem2.clone()))
        m.model.precision = 'fp32'
        assert torch.equal(m.model.decode(tgt, mem2), h0)
        # (3) weights change in place: same memory tensor, stale projection
        m.model.decoder.layers[0].cross_attn.in_proj_weight.mul_(1.25)
        h3 = m.model.decode(tgt, mem2)
        assert torch.equal(h3, m.model.decode(tgt, mem2.clone()))
        assert (h3 - h0).abs().max() > 1e-3                      # and it really is a different result
        # malformed queries are refused with the intended error, not an IndexError
        with pytest.raises(RuntimeError, match='tgt_query must be'):
            m.model.decode(tgt, mem2, tgt_query=torch.zeros(4, device=DEV))
