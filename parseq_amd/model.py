"""Inner PARSeq model: parameter container with the reference's state_dict layout, forward on libparseq_hip.

Mirrors `strhub/models/parseq/model.py:31-169` (class `PARSeq`): same constructor arguments, same attributes
(`decode_ar`, `refine_iters`, `max_label_length`, `encoder`, `decoder`, `head`, `text_embed`, `pos_queries`), same
`forward(tokenizer, images, max_length=None)`, `encode(img)` and `decode(...)` entry points, and exactly the reference's
`state_dict()` keys/shapes so released checkpoints load with `load_state_dict`.

The sub-modules here only HOLD parameters (their own `forward` is never called).  All arithmetic runs in the HIP
library through `parseq_amd._native`; on a CPU tensor or without the library the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from . import _native
from .tokenizer import Tokenizer

_PRECISIONS = {'bf16': _native.PARSEQ_BF16, 'fp32': _native.PARSEQ_F32, 'f32': _native.PARSEQ_F32, 'bf16x3': _native.PARSEQ_BF16X3}


def init_weights(module: nn.Module, name: str = '', exclude: Sequence[str] = ()):
    """Initialisation scheme of the reference (`strhub/models/utils.py:107-125`): trunc-normal(0.02) Linear / Embedding
    weights, zero biases, unit LayerNorm, Kaiming-normal(fan_out) convolutions.  This is a deliberate statement-for-statement
    behavioural mirror of that 19-line function (same branches, same torch initialisers in the same order), not an independent
    design: seeded random-init weights must come out identical to the reference's for bench.py's `manual_seed(0)` model."""
    if any(name.startswith(e) for e in exclude):
        return
    if isinstance(module, nn.Linear):
        nn.init.trunc_normal_(module.weight, std=0.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.trunc_normal_(module.weight, std=0.02)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    elif isinstance(module, nn.Conv2d):
        nn.init.kaiming_normal_(module.weight, mode='fan_out', nonlinearity='relu')
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, (nn.LayerNorm, nn.BatchNorm2d, nn.GroupNorm)):
        nn.init.ones_(module.weight)
        nn.init.zeros_(module.bias)


class _Holder(nn.Module):
    """A module that exists to own parameters under the reference's names."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('parameter holder: compute runs in libparseq_hip, call the PARSeq model instead')


class _PatchEmbed(_Holder):
    def __init__(self, patch_size, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=tuple(patch_size), stride=tuple(patch_size))


class _Attention(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_Holder):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(_Holder):
    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class Encoder(_Holder):
    """Parameter layout of timm's VisionTransformer as the reference configures it (modules.py:128-161):
    no class token, no head; LayerNorm eps 1e-6."""

    def __init__(self, img_size, patch_size, embed_dim, depth, num_heads, mlp_ratio):
        super().__init__()
        self.img_size, self.patch_size = tuple(img_size), tuple(patch_size)
        self.num_heads = num_heads
        n_tok = (img_size[0] // patch_size[0]) * (img_size[1] // patch_size[1])
        self.patch_embed = _PatchEmbed(patch_size, embed_dim)
        self.pos_embed = nn.Parameter(torch.empty(1, n_tok, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        for m in self.modules():      # timm ViT init: trunc-normal(0.02) Linear weights, zero biases; conv keeps torch default
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}


class DecoderLayer(_Holder):
    """Parameter layout of the reference's two-stream pre-LN decoder layer (modules.py:27-52)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout=0.1, layer_norm_eps=1e-5):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.cross_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm_q = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm_c = nn.LayerNorm(d_model, eps=layer_norm_eps)


class Decoder(_Holder):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([DecoderLayer(d_model, nhead, dim_feedforward, dropout) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = nn.LayerNorm(d_model)


class TokenEmbedding(_Holder):
    def __init__(self, charset_size: int, embed_dim: int):
        super().__init__()
        self.embedding = nn.Embedding(charset_size, embed_dim)
        self.embed_dim = embed_dim


def _named_apply(fn, module: nn.Module, name: str = ''):
    for child_name, child in module.named_children():
        full = f'{name}.{child_name}' if name else child_name
        _named_apply(fn, child, full)
        fn(module=child, name=full)


class _Head(nn.Linear):
    """`model.head` (model.py:63): a Linear whose forward runs on the HIP library (fp32 GEMM), so that the reference idiom
    `model.head(model.decode(...))` works; parameters under the reference keys `head.weight` / `head.bias`."""

    def forward(self, x: Tensor) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError('parseq_amd runs on MI355X through libparseq_hip only (no CPU fallback)')
        a = x.detach().to(torch.float32).contiguous().view(-1, self.in_features)
        out = torch.empty(a.shape[0], self.out_features, dtype=torch.float32, device=x.device)
        w = self.weight.detach().to(torch.float32).contiguous()
        b = self.bias.detach().to(torch.float32).contiguous()
        with _native.guard(a):
            _native.check(_native.lib().parseq_op_linear(_native.ptr(a), _native.ptr(w), _native.ptr(b), _native.ptr(out), _native.PARSEQ_F32, 0,
                                                         a.shape[0], self.out_features, self.in_features, _native.stream_ptr(a)))
        return out.view(*x.shape[:-1], self.out_features)


class _NativeState:
    """Device-side twin of the parameters: one parseq_model + cached plans.  Rebuilt when parameters move or change."""

    def __init__(self):
        self.model = C.c_void_p(0)
        self.plans = {}            # (precision code, slot) -> (plan handle, max_batch)
        self.signature = None
        self.param_order = None    # (native model handle, [(state_dict key, numel)] in the library's parameter order)
        self.epoch = 0             # bumped whenever a plan's device-side contents may have changed under an unchanged signature:
                                   # plans created / destroyed / re-packed (re-upload, mark_dirty, an optimiser step on the device)

    def release(self):
        try:
            lib = _native.lib()
        except Exception:
            return
        for plan, _ in self.plans.values():
            lib.parseq_plan_destroy(plan)
        self.plans = {}
        if self.model:
            lib.parseq_model_destroy(self.model)
        self.model = C.c_void_p(0)
        self.signature = None
        self.epoch += 1

    def __del__(self):
        self.release()


class _NativeBacked(nn.Module):
    """A parameter container whose arithmetic lives in libparseq_hip: keeps a device-side twin of the parameters
    (`parseq_model`) and cached workspaces (`parseq_plan`) in step with the module's tensors.  Subclasses provide
    `_cfg` (dict with at least 'img_size'), `precision`, and `_make_native_config()`."""

    @property
    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _make_native_config(self):  # pragma: no cover
        raise NotImplementedError

    def set_profiling(self, enable: bool, batch: int) -> None:
        """Bracket every kernel launch with HIP events (per-family timing for the roofline report; perturbs throughput)."""
        with _native.guard(self._device):
            _native.check(_native.lib().parseq_plan_set_profiling(self._plan(batch), 1 if enable else 0))

    def get_profile(self, batch: int) -> dict:
        """{family: (total_ms, launches)} accumulated since set_profiling(True)."""
        lib, plan, out, i = _native.lib(), self._plan(batch), {}, 0
        while True:
            name, ms, n = C.c_char_p(), C.c_double(), C.c_int64()
            status = lib.parseq_plan_get_profile(plan, i, C.byref(name), C.byref(ms), C.byref(n))
            if status == 1:
                break
            _native.check(status)
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    def _check_images(self, images: Tensor) -> Tensor:
        if not isinstance(images, Tensor) or images.dim() != 4:
            raise RuntimeError('images must be a [N, 3, H, W] tensor')
        if images.device.type != 'cuda':
            raise RuntimeError(
                'parseq_amd runs on MI355X through libparseq_hip only; got a tensor on '
                f"'{images.device}'. There is no CPU fallback (the CPU oracle under oracle/ is test infrastructure).")
        if images.device != self._device:
            raise RuntimeError(f'images on {images.device} but the model is on {self._device}')
        h, w = self._cfg['img_size']
        if tuple(images.shape[1:]) != (3, h, w):
            raise RuntimeError(f'expected images of shape [N, 3, {h}, {w}], got {list(images.shape)}')
        # uint8 = raw pixels: the reference transform's ToTensor + Normalize(0.5, 0.5) (strhub/data/module.py:78-81) is
        # applied inside the patch-embed kernel (row N2); float tensors are taken as already normalised
        if images.dtype not in (torch.float32, torch.bfloat16, torch.uint8):
            images = images.float()
        return images.contiguous()

    @staticmethod
    def _version_of(t: Tensor):
        """In-place-modification counter, or None for tensors that do not keep one (created under torch.inference_mode): an
        in-place edit of such a tensor cannot be noticed, so it is never taken as "unchanged"."""
        return None if t.is_inference() else t._version

    def _signature(self):
        # device, storage addresses and autograd versions of every parameter: load_state_dict, .to(), optimiser steps and
        # in-place ops (also under no_grad) change it; writes through `param.data` do not bump the version and are NOT seen
        params = list(self.parameters())
        return (str(self._device), tuple(p.data_ptr() for p in params), tuple(self._version_of(p) for p in params))

    @staticmethod
    def _trackable(sig) -> bool:
        """False when a parameter keeps no version counter (a tensor created under torch.inference_mode): an in-place edit of it
        could not be noticed, so such a signature never counts as "unchanged" and every call re-uploads (slow, never stale)."""
        return sig is not None and all(v is not None for v in sig[2])

    def mark_dirty(self):
        """Force the next forward / encode / decode to re-upload every parameter.  Needed only after writes the signature cannot
        see: in-place edits through `param.data` (checkpoint averaging, EMA) do not bump a tensor's version counter."""
        self._native_state.signature = None

    def _sync_native(self):
        st: _NativeState = self._native_state
        sig = self._signature()
        if st.signature == sig and self._trackable(sig):
            return st
        if not self._trackable(sig) and not getattr(self, '_warned_untracked', False):
            import warnings
            warnings.warn('parseq_amd: a parameter was created under torch.inference_mode and keeps no version counter; in-place edits of it '
                          'cannot be detected, so every call re-uploads the weights. Create / load the model outside inference_mode.')
            self._warned_untracked = True
        lib = _native.lib()
        if self._device.type != 'cuda':
            raise RuntimeError('move the model to a ROCm device first: model.to("cuda")')
        with torch.cuda.device(self._device):
            fresh = (not st.model) or (st.signature is None) or st.signature[0] != sig[0]
            if fresh:
                st.release()
                cfg = self._make_native_config()
                handle = C.c_void_p(0)
                _native.check(lib.parseq_model_create(C.byref(cfg), C.byref(handle)))
                st.model = handle
            stream = _native.stream_ptr(self._device)
            sd = self.state_dict()
            n_native = lib.parseq_model_num_params(st.model)
            if n_native != len(sd):
                raise RuntimeError(f'state_dict has {len(sd)} tensors, native model expects {n_native}')
            keep = []
            for key, t in sd.items():
                t32 = t.detach().to(dtype=torch.float32).contiguous()
                keep.append(t32)
                _native.check(lib.parseq_model_set_param(st.model, key.encode(), _native.ptr(t32), t32.numel(), stream))
            for plan, _ in st.plans.values():
                _native.check(lib.parseq_plan_refresh(plan, stream))
            torch.cuda.current_stream(self._device).synchronize()    # staging copies in `keep` must outlive the async D2D copies
        st.signature = sig
        st.epoch += 1                     # every plan was re-packed (or released): cached K / V projections are stale
        return st

    def _adopt_native_weights(self):
        """After an optimiser step on the device (`parseq_adamw_step` updates the library's fp32 master weights in place): copy
        them back into this module's parameter tensors — through raw pointers, so their autograd versions, and with them the
        signature, do not change and nothing is uploaded again — and re-pack the plans' bf16 copies / decoder tables."""
        st: _NativeState = self._native_state
        lib, stream = _native.lib(), _native.stream_ptr(self._device)
        sd = self.state_dict()
        n = lib.parseq_model_num_params(st.model)
        if st.param_order is None or st.param_order[0] != st.model.value:      # the library's own parameter order, asked for once per native model
            order = []
            for i in range(n):
                key, numel = C.c_char_p(), C.c_int64()
                _native.check(lib.parseq_model_param_info(st.model, i, C.byref(key), C.byref(numel)))
                order.append((key.value.decode(), numel.value))
            st.param_order = (st.model.value, order)
        ptrs = (C.c_void_p * n)()
        for i, (key, numel) in enumerate(st.param_order[1]):
            t = sd[key]
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != numel:
                raise RuntimeError(f'parameter {key}: training needs contiguous fp32 parameters of the model\'s shape')
            ptrs[i] = t.data_ptr()
        _native.check(lib.parseq_model_get_params(st.model, ptrs, n, stream))      # one launch for all of them
        for plan, _ in st.plans.values():
            _native.check(lib.parseq_plan_refresh(plan, stream))
        st.epoch += 1                     # same signature, new weights: a cached memory K / V projection is stale

    def _plan(self, batch: int, slot: int = 0):
        if self.precision not in _PRECISIONS:
            raise RuntimeError(f"precision must be one of {sorted(_PRECISIONS)}, got '{self.precision}'")
        st = self._sync_native()
        code = _PRECISIONS[self.precision]
        plan, cap = st.plans.get((code, slot), (None, 0))
        if plan is None or batch > cap:
            lib = _native.lib()
            if plan is not None:
                torch.cuda.current_stream(self._device).synchronize()
                lib.parseq_plan_destroy(plan)
                del st.plans[(code, slot)]
            st.epoch += 1                 # a fresh plan holds nobody's K / V (and a freed handle's address may be reused)
            cap = max(8, 1 << (batch - 1).bit_length())
            handle = C.c_void_p(0)
            with torch.cuda.device(self._device):
                if os.environ.get('PARSEQ_PLAN_ALLOCATOR', 'torch') == 'torch':
                    # the plan's arena is a block of the CALLER's allocator — torch's caching allocator (parseq_plan_create_ex; SURVEY.md section 8(b)'s
                    # ownership clause) — the default since round 6; PARSEQ_PLAN_ALLOCATOR=hip opts out to a hipMalloc outside it
                    if getattr(st, 'allocator', None) is None:
                        st.allocator = _native.TorchPlanAllocator(self._device)
                    _native.check(lib.parseq_plan_create_ex(st.model, cap, code, _native.stream_ptr(self._device), st.allocator.alloc_ptr, st.allocator.release_ptr,
                                                            None, C.byref(handle)))
                else:
                    _native.check(lib.parseq_plan_create(st.model, cap, code, _native.stream_ptr(self._device), C.byref(handle)))
            st.plans[(code, slot)] = (handle, cap)
            plan = handle
        return plan

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        st = getattr(self, '_native_state', None)
        if st is not None:
            st.signature = None if st.signature is None else (st.signature[0], (), ())   # force a re-sync
        return out


class PARSeq(_NativeBacked):

    def __init__(self, num_tokens: int, max_label_length: int, img_size: Sequence[int], patch_size: Sequence[int],
                 embed_dim: int, enc_num_heads: int, enc_mlp_ratio: int, enc_depth: int, dec_num_heads: int,
                 dec_mlp_ratio: int, dec_depth: int, decode_ar: bool, refine_iters: int, dropout: float,
                 precision: Optional[str] = None) -> None:
        super().__init__()
        self.max_label_length = max_label_length
        self.decode_ar = decode_ar
        self.refine_iters = refine_iters
        self.precision = precision or os.environ.get('PARSEQ_AMD_PRECISION', 'bf16x3')      # the mode that meets the reference within 1e-3; 'bf16' = throughput mode
        self._cfg = dict(num_tokens=num_tokens, img_size=tuple(img_size), patch_size=tuple(patch_size), embed_dim=embed_dim,
                         enc_num_heads=enc_num_heads, enc_mlp_ratio=enc_mlp_ratio, enc_depth=enc_depth,
                         dec_num_heads=dec_num_heads, dec_mlp_ratio=dec_mlp_ratio, dec_depth=dec_depth)

        self.encoder = Encoder(img_size, patch_size, embed_dim, enc_depth, enc_num_heads, enc_mlp_ratio)
        self.decoder = Decoder(embed_dim, dec_num_heads, embed_dim * dec_mlp_ratio, dropout, dec_depth)
        self.head = _Head(embed_dim, num_tokens - 2)           # <bos> and <pad> are never predicted (model.py:62-63)
        self.text_embed = TokenEmbedding(num_tokens, embed_dim)
        self.pos_queries = nn.Parameter(torch.empty(1, max_label_length + 1, embed_dim))
        _named_apply(lambda module, name: init_weights(module, name, exclude=['encoder']), self)
        nn.init.trunc_normal_(self.pos_queries, std=0.02)
        object.__setattr__(self, '_native_state', _NativeState())

    # ---- reference surface -------------------------------------------------------------------------------------
    @property
    def _device(self) -> torch.device:
        return self.head.weight.device

    def no_weight_decay(self):
        return {'text_embed.embedding.weight', 'pos_queries'} | {'encoder.' + n for n in self.encoder.no_weight_decay()}

    def encode(self, img: Tensor) -> Tensor:
        """images [B, 3, H, W] -> memory [B, tokens, E] (fp32).  model.py:83-84."""
        img = self._check_images(img)
        plan = self._plan(img.shape[0])
        memory = torch.empty(img.shape[0], self.encoder.pos_embed.shape[1], self._cfg['embed_dim'],
                             dtype=torch.float32, device=img.device)
        _native.check(_native.lib().parseq_encode(plan, _native.ptr(img), _native.dtype_code(img.dtype), img.shape[0],
                                                  _native.ptr(memory), _native.stream_ptr(self._device)))
        self._remember_memory(memory, plan)      # the K / V projection of THIS tensor object is what THIS plan now caches
        return memory

    def _bind_memory(self, memory: Optional[Tensor], B: int, plan) -> None:
        """Make the plan's cross-attention K / V those of `memory` (model.py:89).  `None`, or the very tensor the last
        `encode` on this model returned (same storage, unmodified since), means the cached projection is already right;
        anything else — another batch's memory, an edited tensor, a memory produced elsewhere — is re-projected
        (`parseq_set_memory`), never silently ignored."""
        if memory is None:
            return
        E = self._cfg['embed_dim']
        n_tok = self.encoder.pos_embed.shape[1]
        if memory.dim() != 3 or tuple(memory.shape) != (B, n_tok, E):
            raise RuntimeError(f'memory must be [{B}, {n_tok}, {E}], got {list(memory.shape)}')
        if memory.device != self._device:
            raise RuntimeError(f'memory on {memory.device} but the model is on {self._device}')
        key = getattr(self, '_memory_key', None)
        if key is not None and key[0]() is memory and key[1] is not None and key[1:] == (self._version_of(memory), plan.value, self._native_state.epoch):
            return
        mem = memory.detach().to(torch.float32).contiguous()
        _native.check(_native.lib().parseq_set_memory(plan, _native.ptr(mem), B, _native.stream_ptr(self._device)))
        self._memory_keep = mem                 # stays alive until the asynchronous projection has read it
        self._remember_memory(memory, plan)

    def _remember_memory(self, memory: Tensor, plan) -> None:
        # identity of the tensor OBJECT (a weak reference: a freed tensor whose storage address is reused by another one can
        # never match), its in-place-modification counter, the PLAN that holds the projection (another precision or slot is another
        # plan with its own K / V) and the state epoch (bumped by every re-upload, re-pack, optimiser step and plan creation)
        import weakref
        self._memory_key = (weakref.ref(memory), self._version_of(memory), plan.value, self._native_state.epoch)


    def decode(self, tgt: Tensor, memory: Optional[Tensor] = None, tgt_mask: Optional[Tensor] = None,
               tgt_padding_mask: Optional[Tensor] = None, tgt_query: Optional[Tensor] = None,
               tgt_query_mask: Optional[Tensor] = None) -> Tensor:
        """model.py:86-103: decoder output (after decoder.norm) [N, Lq, E] for the context tokens `tgt`.

        `memory`: the encoder output to attend to.  The tensor returned by the most recent `encode` is recognised (its K / V
        projection is still cached on the device); any other tensor is projected afresh; `None` means "the most recent
        `encode` / `forward`".  `tgt_query`: None (all positions `pos_queries[:, :L]`), a view `pos_queries[:, i:j]` (what
        `forward` and the training step pass — served from the position-query tables), or any other [N, Lq, E] / [1, Lq, E]
        tensor (projected at call time).  `tgt_mask` only feeds the content-stream update, which a depth-1 decoder never
        runs (modules.py:116-118), so it is ignored exactly as the reference ignores it."""
        B, L = tgt.shape
        E = self._cfg['embed_dim']
        user_query = None
        if tgt_query is None:
            q_start, q_len = 0, L
        else:
            pq = self.pos_queries
            if not isinstance(tgt_query, Tensor) or tgt_query.dim() != 3 or tgt_query.shape[-1] != E:
                raise RuntimeError(f'tgt_query must be [{B} or 1, Lq, {E}], got {list(getattr(tgt_query, "shape", []))}')
            off = (tgt_query.data_ptr() - pq.data_ptr()) // pq.element_size()
            q_len = tgt_query.shape[1]
            is_slice = (tgt_query.device == pq.device and tgt_query.dtype == pq.dtype and 0 <= off and off % E == 0 and
                        tgt_query.dim() == 3 and tgt_query.shape[-1] == E and off // E + q_len <= pq.shape[1] and
                        tgt_query.stride(-1) == 1 and tgt_query.stride(-2) == E and
                        (tgt_query.shape[0] == 1 or tgt_query.stride(0) == 0))
            if is_slice:
                q_start = off // E
            else:
                if tgt_query.dim() != 3 or tgt_query.shape[-1] != E or tgt_query.shape[0] not in (1, B):
                    raise RuntimeError(f'tgt_query must be [{B} or 1, Lq, {E}], got {list(tgt_query.shape)}')
                q_start = 0
                user_query = tgt_query.detach().to(device=self._device, dtype=torch.float32).expand(B, q_len, E).contiguous()
        plan = self._plan(B)
        self._bind_memory(memory, B, plan)
        tok = tgt.to(device=self._device, dtype=torch.int32).contiguous()
        kpm = tgt_padding_mask.to(device=self._device, dtype=torch.uint8).contiguous() if tgt_padding_mask is not None else None
        qm = tgt_query_mask.to(device=self._device, dtype=torch.uint8).contiguous() if tgt_query_mask is not None else None
        if qm is not None and tuple(qm.shape) != (q_len, L):
            raise RuntimeError(f'tgt_query_mask shape {tuple(qm.shape)} != ({q_len}, {L})')
        hidden = torch.empty(B, q_len, E, dtype=torch.float32, device=self._device)
        logits = torch.empty(B, q_len, self._cfg['num_tokens'] - 2, dtype=torch.float32, device=self._device)
        lib, stream = _native.lib(), _native.stream_ptr(self._device)
        if user_query is None:
            _native.check(lib.parseq_decode_hidden(plan, _native.ptr(tok), B, L, q_start, q_len, _native.ptr(qm), _native.ptr(kpm),
                                                   _native.ptr(hidden), _native.ptr(logits), stream))
        else:
            _native.check(lib.parseq_decode_query(plan, _native.ptr(tok), B, L, _native.ptr(user_query), q_len, _native.ptr(qm),
                                                  _native.ptr(kpm), _native.ptr(hidden), _native.ptr(logits), stream))
        return hidden

    def decode_logits(self, tgt: Tensor, q_start: int, q_len: int, tgt_padding_mask: Optional[Tensor] = None,
                      tgt_query_mask: Optional[Tensor] = None) -> Tensor:
        """head(decode(...)) for queries pos_queries[q_start:q_start+q_len] against the content tokens `tgt`
        (model.py:86-103 + :138), using the memory of the most recent `encode` / `forward` on this model.
        Per-stage parity hook; masks use torch's convention (True = masked)."""
        B, L = tgt.shape
        plan = self._plan(B)
        tok = tgt.to(device=self._device, dtype=torch.int32).contiguous()
        kpm = tgt_padding_mask.to(device=self._device, dtype=torch.uint8).contiguous() if tgt_padding_mask is not None else None
        qm = tgt_query_mask.to(device=self._device, dtype=torch.uint8).contiguous() if tgt_query_mask is not None else None
        if qm is not None and tuple(qm.shape) != (q_len, L):
            raise RuntimeError(f'tgt_query_mask shape {tuple(qm.shape)} != ({q_len}, {L})')
        out = torch.empty(B, q_len, self._cfg['num_tokens'] - 2, dtype=torch.float32, device=self._device)
        _native.check(_native.lib().parseq_decode_logits(plan, _native.ptr(tok), B, L, q_start, q_len, _native.ptr(qm),
                                                         _native.ptr(kpm), _native.ptr(out), _native.stream_ptr(self._device)))
        return out

    def forward(self, tokenizer: Tokenizer, images: Tensor, max_length: Optional[int] = None, slot: Optional[int] = None,
                return_length: bool = False):
        """model.py:105-169.  Returns logits [B, L, num_tokens - 2] (fp32).

        `return_length=True` returns `(logits_all, L)` instead: the logits of ALL `num_steps` positions (every step is always
        run on the device) and the length `L` the reference would have returned for THIS batch — what the data-parallel wrapper
        needs to reproduce the single-device early-exit length across shards (parallel.data_parallel_forward).

        `slot` selects one of several independent workspaces (plans): calls that use different slots on different
        streams may be in flight at the same time (bench.py --streams 2 overlaps the latency-bound AR decode of one batch
        with the encoder of the next).  The default (`slot=None`) is the reference's behaviour: one call at a time,
        stream-ordered, on workspace 0 — and it tells the library so (PARSEQ_FLAG_LATENCY): with the device to itself the AR step
        spreads over more compute units, a shorter dependent chain.  An explicit slot (0 included) says other batches are in flight
        and keeps the narrower step, whose compute units they need; the two forms differ by rounding only."""
        testing = max_length is None
        max_length = self.max_label_length if max_length is None else min(max_length, self.max_label_length)
        num_steps = max_length + 1
        images = self._check_images(images)
        B = images.shape[0]
        plan = self._plan(B, 0 if slot is None else slot)
        key = getattr(self, '_memory_key', None)
        if key is not None and key[2] == plan.value:
            self._memory_key = None             # this plan's cached K / V now belong to these images, not to an encode() result
        if (tokenizer.bos_id, tokenizer.eos_id, tokenizer.pad_id) != self._special_ids(tokenizer):
            raise RuntimeError('tokenizer special ids changed after the native model was built')
        logits = torch.empty(B, num_steps, self._cfg['num_tokens'] - 2, dtype=torch.float32, device=images.device)
        flags = ((_native.FLAG_DECODE_AR if self.decode_ar else 0) | (_native.FLAG_TESTING if testing else 0)
                 | (_native.FLAG_LATENCY if slot is None else 0))
        out_len = C.c_int(0)
        _native.check(_native.lib().parseq_forward(plan, _native.ptr(images), _native.dtype_code(images.dtype), B, flags,
                                                   int(self.refine_iters), num_steps, _native.ptr(logits),
                                                   C.byref(out_len), _native.stream_ptr(self._device)))
        if return_length:
            return logits, out_len.value
        return logits if out_len.value == num_steps else logits[:, :out_len.value]

    # ---- native plumbing ---------------------------------------------------------------------------------------
    def _special_ids(self, tokenizer):
        self._tok_ids = getattr(self, '_tok_ids', None) or (tokenizer.bos_id, tokenizer.eos_id, tokenizer.pad_id)
        return self._tok_ids

    def _make_native_config(self):
        tok_ids = getattr(self, '_tok_ids', None)
        if tok_ids is None:
            n = self._cfg['num_tokens']
            tok_ids = (n - 2, 0, n - 1)      # Tokenizer layout: [E]=0 ... [B]=n-2, [P]=n-1 (strhub/data/utils.py:107-111)
            self._tok_ids = tok_ids
        c = self._cfg
        return _native.ParseqConfig(
            img_h=c['img_size'][0], img_w=c['img_size'][1], patch_h=c['patch_size'][0], patch_w=c['patch_size'][1],
            embed_dim=c['embed_dim'], enc_depth=c['enc_depth'], enc_heads=c['enc_num_heads'],
            enc_mlp_ratio=c['enc_mlp_ratio'], dec_depth=c['dec_depth'], dec_heads=c['dec_num_heads'],
            dec_mlp_ratio=c['dec_mlp_ratio'], num_tokens=c['num_tokens'], max_label_length=self.max_label_length,
            bos_id=tok_ids[0], eos_id=tok_ids[1], pad_id=tok_ids[2], enc_ln_eps=1e-6, dec_ln_eps=1e-5,
            arch=_native.ARCH_PARSEQ)
