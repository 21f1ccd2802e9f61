"""CPU tests: the oracle (oracle/parseq_oracle.py) against golden vectors minted by executing the reference's
own model.py / modules.py (oracle/make_golden.py), plus properties of the reference algorithm established in
SURVEY.md section 4 and an independent second opinion on the ViT block conventions."""
import pytest
import torch

from oracle import parseq_oracle as O
from oracle.synth import CONFIGS, state_dict_fingerprint, state_dict_spec, synth_images, synth_state_dict

MODES = {  # name: (decode_ar, refine_iters, max_length) — mirrors oracle/make_golden.py
    'nar0': (False, 0, None), 'nar1': (False, 1, None), 'ar0': (True, 0, None), 'ar0_full': (True, 0, 25),
    'ar0_len7': (True, 0, 7), 'ar1': (True, 1, None), 'ar2': (True, 2, None),
}


@pytest.fixture(scope='module', params=['parseq', 'parseq-tiny', 'parseq-patch16-224'])
def setup(request, golden):
    name = request.param
    cfg = CONFIGS[name]
    sd = synth_state_dict(cfg, 0)
    g, meta = golden(name)
    return name, cfg, sd, g, meta


def test_synth_weights_reproduce(setup):
    """The per-key seeded generator gives the same bits here as where the goldens were minted."""
    _, cfg, sd, g, meta = setup
    assert state_dict_fingerprint(sd) == meta['sd_fingerprint']
    assert sum(v.numel() for v in sd.values()) == meta['num_params']


def test_param_counts_match_readme():
    """README.md:220-226 of the reference (PARSeq-S) and SURVEY section 4 (Ti): weight-independent known answers."""
    n = {k: sum(int(torch.tensor(s).prod()) for s in state_dict_spec(c).values()) for k, c in CONFIGS.items()}
    assert n['parseq'] == 23_832_671 and n['parseq-tiny'] == 6_018_143
    spec = state_dict_spec(CONFIGS['parseq'])
    part = lambda pre: sum(int(torch.tensor(s).prod()) for k, s in spec.items() if k.startswith(pre))
    assert part('encoder.') == 21_380_736 and part('decoder.') == 2_368_128
    assert part('head.') == 36_575 and part('text_embed.') == 37_248 and part('pos_queries') == 9_984


def test_encoder_matches_reference(setup):
    _, cfg, sd, g, _ = setup
    with torch.inference_mode():
        mem = O.encode(sd, cfg, g['images'])
    torch.testing.assert_close(mem, g['memory'], rtol=0, atol=2e-6)


@pytest.mark.parametrize('mode', list(MODES))
def test_forward_matches_reference(setup, mode):
    """Every decode mode: same shape (early exit / max_length rules), same logits, same argmax."""
    _, cfg, sd, g, meta = setup
    ar, ri, ml = MODES[mode]
    with torch.inference_mode():
        lo = O.forward(sd, cfg, g['images'], ml, decode_ar=ar, refine_iters=ri)
    ref = g[f'logits.{mode}']
    assert list(lo.shape) == meta['modes'][mode]['shape'] == list(ref.shape)
    torch.testing.assert_close(lo, ref, rtol=0, atol=5e-6)
    assert torch.equal(lo.argmax(-1), ref.argmax(-1))


def test_output_length_rules(setup):
    """SURVEY section 4 item 3 / README.md:111-112: L<=26 by early exit; max_length=k -> min(k,25)+1; refine -> 26."""
    _, cfg, sd, g, meta = setup
    assert meta['modes']['ar0']['shape'][1] < 26          # the synthetic weights do trigger early exit
    assert meta['modes']['ar0_full']['shape'][1] == 26
    assert meta['modes']['ar0_len7']['shape'][1] == 8
    assert meta['modes']['ar1']['shape'] == [g['images'].shape[0], 26, 95]
    with torch.inference_mode():
        lo = O.forward(sd, cfg, g['images'][:1], 99, decode_ar=True, refine_iters=0)
    assert lo.shape == (1, 26, 95)


def test_batch_invariance(setup):
    _, cfg, sd, g, _ = setup
    with torch.inference_mode():
        lo = O.forward(sd, cfg, g['images'][:1], None, decode_ar=True, refine_iters=1)
    torch.testing.assert_close(lo, g['logits.ar1.batch1'], rtol=0, atol=5e-6)
    torch.testing.assert_close(lo[0], g['logits.ar1'][0], rtol=0, atol=2e-5)


def test_ar_loop_equals_teacher_forced_single_pass(setup):
    """the step-by-step AR loop equals one pass over the full context with the causal mask."""
    _, cfg, sd, g, _ = setup
    images = g['images'][:4]
    with torch.inference_mode():
        tr = O.Trace()
        lo = O.forward(sd, cfg, images, 25, decode_ar=True, refine_iters=0, trace=tr)
        causal = torch.triu(torch.ones(26, 26, dtype=torch.bool), 1)
        pos_q = sd['pos_queries'].expand(images.shape[0], -1, -1)
        one = O.head(sd, O.decode(sd, cfg, tr.ar_tokens, tr.memory, causal, None, pos_q, causal))
    torch.testing.assert_close(one, lo, rtol=0, atol=2e-5)


def test_early_exit_does_not_change_refinement(setup):
    """The HIP path always runs all 26 AR steps before refinement; positions past the all-EOS step are behind the
    padding mask, so refinement output must not depend on where the AR loop stopped."""
    _, cfg, sd, g, _ = setup
    with torch.inference_mode():
        tr = O.Trace()
        O.forward(sd, cfg, g['images'], 25, decode_ar=True, refine_iters=0, trace=tr)       # 26 steps, no exit
        full_tokens = tr.ar_tokens
        bos_shift = full_tokens                                                              # [bos, tok_1..tok_25]
        lo = O.forward(sd, cfg, g['images'], None, decode_ar=True, refine_iters=1, teacher_refine_tokens=[bos_shift])
    torch.testing.assert_close(lo, g['logits.ar1'], rtol=0, atol=2e-5)


def test_bf16_rounding_oracle_is_close_and_argmax_identical(setup):
    """Sizes the honest gap between the throughput mode's arithmetic and exact fp32 on these goldens."""
    _, cfg, sd, g, _ = setup
    with torch.inference_mode():
        lb = O.forward(sd, cfg, g['images'], None, decode_ar=True, refine_iters=1, rounding='bf16')
    assert (lb - g['logits.ar1']).abs().max() < 0.1
    assert torch.equal(lb.argmax(-1), g['logits.ar1'].argmax(-1))


def test_vit_block_against_transformers_vitlayer():
    """Second opinion on the un-vendored timm block semantics (pre-LN, eps, erf-GELU, fused-qkv row order,
    softmax scale): an independent implementation (HF transformers ViTLayer) with the same weights."""
    mv = pytest.importorskip('transformers.models.vit.modeling_vit')
    cfg = O.OracleConfig(enc_depth=1)
    sd = synth_state_dict(cfg, 3)
    E = cfg.embed_dim
    hf_cfg = mv.ViTConfig(hidden_size=E, num_attention_heads=cfg.enc_num_heads, intermediate_size=4 * E,
                          hidden_act='gelu', layer_norm_eps=1e-6, qkv_bias=True, hidden_dropout_prob=0.0,
                          attention_probs_dropout_prob=0.0)
    hf_cfg._attn_implementation = 'eager'
    layer = mv.ViTLayer(hf_cfg).eval()
    p = 'encoder.blocks.0.'
    w, b = sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias']
    m = {'layernorm_before.weight': sd[p + 'norm1.weight'], 'layernorm_before.bias': sd[p + 'norm1.bias'],
         'layernorm_after.weight': sd[p + 'norm2.weight'], 'layernorm_after.bias': sd[p + 'norm2.bias'],
         'attention.q_proj.weight': w[:E], 'attention.k_proj.weight': w[E:2 * E], 'attention.v_proj.weight': w[2 * E:],
         'attention.q_proj.bias': b[:E], 'attention.k_proj.bias': b[E:2 * E], 'attention.v_proj.bias': b[2 * E:],
         'attention.o_proj.weight': sd[p + 'attn.proj.weight'], 'attention.o_proj.bias': sd[p + 'attn.proj.bias'],
         'mlp.fc1.weight': sd[p + 'mlp.fc1.weight'], 'mlp.fc1.bias': sd[p + 'mlp.fc1.bias'],
         'mlp.fc2.weight': sd[p + 'mlp.fc2.weight'], 'mlp.fc2.bias': sd[p + 'mlp.fc2.bias']}
    layer.load_state_dict(m, strict=True)
    x = torch.randn(2, cfg.num_patches, E, generator=torch.Generator().manual_seed(5))
    with torch.inference_mode():
        want = layer(x)
        want = want[0] if isinstance(want, tuple) else want
        # oracle block = encode() minus patch embed / final norm: run it through a 1-layer config by hand
        h = O._ln(x, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-6)
        qkv = O._linear(h, w, b, None).reshape(2, -1, 3, cfg.enc_num_heads, E // cfg.enc_num_heads).permute(2, 0, 3, 1, 4)
        a = torch.nn.functional.scaled_dot_product_attention(*qkv.unbind(0)).transpose(1, 2).reshape(2, -1, E)
        y = x + O._linear(a, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'], None)
        h = O._ln(y, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-6)
        y = y + O._linear(torch.nn.functional.gelu(O._linear(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'], None)),
                          sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'], None)
    torch.testing.assert_close(y, want, rtol=0, atol=2e-5)


def test_synth_images_range():
    x = synth_images(4, CONFIGS['parseq'])
    assert x.shape == (4, 3, 32, 128) and x.min() >= -1 and x.max() <= 1


def test_whole_vit_encoder_against_transformers_vitmodel():
    """The timm boundary, pinned as far as an offline box allows: the WHOLE encoder the stand-in restates — Conv2d patch embedding with (4, 8)
    patches of a 32 x 128 crop, class token, position embedding added after the projection, twelve pre-LN blocks (eps 1e-6, exact GELU,
    fused-qkv row order, 1 / sqrt(64) soft-max scale), final LayerNorm — against an INDEPENDENT implementation of the same architecture
    (HF transformers `ViTModel`, which like timm's ViT with `class_token=True` prepends a class token: the ViTSTR encoder,
    strhub/models/vitstr/model.py:14-28) with the same synthetic weights, on the crops of the ViTSTR goldens.  `parseq_oracle.vit_features`
    is the ONE function both oracles' encoders run (PARSeq's without the class-token row), so this covers every operation of the PARSeq
    encoder as well.  Not a reference-held vector — timm itself is unobtainable here — but a second opinion on every line of Appendix A."""
    mv = pytest.importorskip('transformers.models.vit.modeling_vit')
    import os
    from safetensors.torch import load_file
    from oracle import vitstr_oracle as V
    cfg = V.vitstr_config()
    sd = V.synth_state_dict(cfg, 0)
    E, H, depth = cfg.embed_dim, cfg.enc_num_heads, cfg.enc_depth
    hf_cfg = mv.ViTConfig(hidden_size=E, num_hidden_layers=depth, num_attention_heads=H, intermediate_size=4 * E, hidden_act='gelu',
                          layer_norm_eps=1e-6, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                          image_size=tuple(cfg.img_size), patch_size=tuple(cfg.patch_size), num_channels=3)
    hf_cfg._attn_implementation = 'eager'
    model = mv.ViTModel(hf_cfg, add_pooling_layer=False).eval()
    m = {'embeddings.cls_token': sd['cls_token'], 'embeddings.position_embeddings': sd['pos_embed'],
         'embeddings.patch_embeddings.projection.weight': sd['patch_embed.proj.weight'],
         'embeddings.patch_embeddings.projection.bias': sd['patch_embed.proj.bias'],
         'layernorm.weight': sd['norm.weight'], 'layernorm.bias': sd['norm.bias']}
    for i in range(depth):
        p, q = f'blocks.{i}.', f'layers.{i}.'
        w, b = sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias']
        m.update({q + 'layernorm_before.weight': sd[p + 'norm1.weight'], q + 'layernorm_before.bias': sd[p + 'norm1.bias'],
                  q + 'layernorm_after.weight': sd[p + 'norm2.weight'], q + 'layernorm_after.bias': sd[p + 'norm2.bias'],
                  q + 'attention.q_proj.weight': w[:E], q + 'attention.k_proj.weight': w[E:2 * E], q + 'attention.v_proj.weight': w[2 * E:],
                  q + 'attention.q_proj.bias': b[:E], q + 'attention.k_proj.bias': b[E:2 * E], q + 'attention.v_proj.bias': b[2 * E:],
                  q + 'attention.o_proj.weight': sd[p + 'attn.proj.weight'], q + 'attention.o_proj.bias': sd[p + 'attn.proj.bias'],
                  q + 'mlp.fc1.weight': sd[p + 'mlp.fc1.weight'], q + 'mlp.fc1.bias': sd[p + 'mlp.fc1.bias'],
                  q + 'mlp.fc2.weight': sd[p + 'mlp.fc2.weight'], q + 'mlp.fc2.bias': sd[p + 'mlp.fc2.bias']})
    res = model.load_state_dict(m, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    images = load_file(os.path.join(os.path.dirname(__file__), 'golden', 'vitstr.safetensors'))['images']
    with torch.inference_mode():
        want = model(images).last_hidden_state                      # [B, 129, E]
        got = O.vit_features(sd, '', cfg, images)
    assert got.shape == want.shape == (images.shape[0], cfg.num_patches + 1, E)
    err = (got - want).abs().max().item()
    print(f'[oracle encoder vs transformers ViTModel] tokens {tuple(got.shape)} max|d| {err:.3e}, |want| max {want.abs().max().item():.3e}')
    assert err <= 5e-5          # twelve blocks of fp32 reassociation on O(1) activations (measured 1e-5)
