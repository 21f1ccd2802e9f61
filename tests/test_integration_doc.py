"""INTEGRATION.md's reference-side binding must stay in step with include/parseq_hip.h.

CPU: the `_Cfg` ctypes structure of the documented shim has exactly the fields of `struct parseq_config`, in order, with the
same C types.  GPU: the documented code block is executed VERBATIM (only the library name is replaced by the in-tree path)
against a module that has the reference's attribute layout (`strhub/models/parseq/model.py:31-81` on timm's ViT) and the
golden state dict, and its forward must reproduce the reference's logits."""
import ctypes as C
import os
import re
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _doc_block():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    block = next(b for b in blocks if 'class HipPARSeq' in b)
    return block


def _header_fields():
    text = open(os.path.join(ROOT, 'include', 'parseq_hip.h')).read()
    body = re.search(r'typedef struct parseq_config \{(.*?)\} parseq_config;', text, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        for n in names.split(','):
            fields.append((n.strip(), ctype))
    return fields


def test_documented_cfg_matches_the_header():
    from parseq_amd import _native
    ns = {}
    block = _doc_block().replace("C.CDLL('libparseq_hip.so')", f"C.CDLL({_native.lib_path()!r})")
    exec(compile(block, 'INTEGRATION.md', 'exec'), ns)
    want = _header_fields()
    ctypes_of = {'int32_t': C.c_int32, 'float': C.c_float}
    got = ns['_Cfg']._fields_
    assert [n for n, _ in got] == [n for n, _ in want]
    assert all(t is ctypes_of[ct] for (_, t), (_, ct) in zip(got, want))
    assert C.sizeof(ns['_Cfg']) == C.sizeof(_native.ParseqConfig)
    assert [n for n, _ in _native.ParseqConfig._fields_] == [n for n, _ in want]       # this repository's own binding too


def _reference_shaped(sd, cfg, decode_ar, refine_iters):
    """An object with the attributes the documented shim reads from a strhub PARSeq (model.py:52-67; timm's PatchEmbed /
    Attention; nn.MultiheadAttention) — none of which exists on the GPU box — and the golden state dict."""
    ns = types.SimpleNamespace
    blocks = [ns(attn=ns(num_heads=cfg.enc_num_heads)) for _ in range(cfg.enc_depth)]
    enc = ns(patch_embed=ns(img_size=tuple(cfg.img_size), patch_size=tuple(cfg.patch_size)), embed_dim=cfg.embed_dim, blocks=blocks)
    dec = ns(layers=[ns(self_attn=ns(num_heads=cfg.dec_num_heads))])
    return ns(encoder=enc, decoder=dec, max_label_length=cfg.max_label_length, decode_ar=decode_ar, refine_iters=refine_iters,
              head=ns(out_features=cfg.num_tokens - 2), state_dict=lambda: sd)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['parseq', 'parseq-tiny'])
def test_documented_binding_runs_verbatim(name, golden):
    from oracle.synth import CONFIGS, synth_state_dict
    from parseq_amd import _native
    from parseq_amd.tokenizer import Tokenizer
    from parseq_amd.configs import CHARSET_94_FULL
    g, meta = golden(name)
    cfg = CONFIGS[name]
    sd = {k: v.cuda() for k, v in synth_state_dict(cfg, 0).items()}
    ns = {}
    block = _doc_block().replace("C.CDLL('libparseq_hip.so')", f"C.CDLL({_native.lib_path()!r})")
    exec(compile(block, 'INTEGRATION.md', 'exec'), ns)
    ns['_lib'].parseq_last_error.restype = C.c_char_p
    tok = Tokenizer(CHARSET_94_FULL)
    images = g['images'].cuda()
    for mode, (ar, refine, max_length) in {'ar1': (True, 1, None), 'nar0': (False, 0, None), 'ar0_full': (True, 0, 25), 'ar0': (True, 0, None)}.items():
        model = _reference_shaped(sd, cfg, ar, refine)
        hip = ns['HipPARSeq'](model, tok, max_batch=8, bf16=False)
        logits = hip.forward(model, images, max_length)
        want = g[f'logits.{mode}']
        assert logits.shape == want.shape, (mode, logits.shape, want.shape)
        assert (logits.cpu() - want).abs().max() <= 1e-3, mode
        ns['_lib'].parseq_plan_destroy(hip.plan)
        ns['_lib'].parseq_model_destroy(hip.model)
