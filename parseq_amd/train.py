"""Row N3, backward: what `loss.backward()` leaves behind after `PARSeq.training_step` (strhub/models/parseq/system.py:168-199).

`loss_and_grads` = the whole step's forward and backward in fp32 from the model's master weights, dropout off:
`parseq_train_encoder_forward` (keeps the activations the backward needs) -> `parseq_train_decoder` (loss, gradient of every
`decoder.*` / `head.*` / `text_embed.*` parameter and `pos_queries`, and d loss / d memory) -> `parseq_train_encoder_backward`
(gradient of every `encoder.*` parameter).  `decoder_backward` is the middle stage alone, on any `memory`.
Not built: dropout, optimiser, gradient all-reduce.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
from torch import Tensor

from . import _native
from .system import generate_attn_masks


@dataclass
class DecoderBackward:
    loss: Tensor                    # [] fp32 on the device
    perm_losses: Tensor             # [K]
    grads: Dict[str, Tensor]        # reference state_dict key -> gradient, views into `flat` (encoder.* are zero)
    flat: Tensor                    # [parseq_model_grad_elems] the buffer as the library lays it out
    dmemory: Tensor                 # [B, tokens, E]
    perms: Tensor
    workspace: Tensor               # raw workspace (floats); `intermediate(name)` reads it
    memory: Optional[Tensor] = None # the training forward's encoder output (loss_and_grads only)
    _shape: tuple = ()
    _model: Optional[object] = None

    def intermediate(self, name: str, numel: int) -> Tensor:
        """A named intermediate of the last permutation / an accumulator (parseq_train_decoder_workspace_offset), flat."""
        B, L, K = self._shape
        off = _native.lib().parseq_train_decoder_workspace_offset(self._model, B, L, K, name.encode())
        if off < 0:
            raise KeyError(name)
        return self.workspace[off:off + numel]


def param_views(native_model, flat: Tensor, shapes: Dict[str, Sequence[int]]) -> Dict[str, Tensor]:
    lib = _native.lib()
    out = {}
    for i in range(lib.parseq_model_num_params(native_model)):
        key, numel = C.c_char_p(), C.c_int64()
        _native.check(lib.parseq_model_param_info(native_model, i, C.byref(key), C.byref(numel)))
        off = lib.parseq_model_param_offset(native_model, i)
        name = key.value.decode()
        out[name] = flat[off:off + numel.value].view(*shapes[name])
    return out


def decoder_backward(system, images: Tensor, labels, perms: Optional[Tensor] = None, memory: Optional[Tensor] = None) -> DecoderBackward:
    """One training batch up to and including the decoder's backward.  `perms` defaults to a fresh draw from the system's
    sampler (system.py:175); `memory` defaults to `system.model.encode(images)` in the system's precision."""
    lib = _native.lib()
    model = system.model
    dev = system.device
    tgt = system.tokenizer.encode(labels, dev)
    if perms is None:
        perms = system.gen_tgt_perms(tgt)
    tgt_in, tgt_out = tgt[:, :-1], tgt[:, 1:]
    B, L = tgt_in.shape
    K = len(perms)
    late = torch.where(tgt_out == system.eos_id, system.pad_id, tgt_out)
    targets = torch.stack([tgt_out.reshape(-1), late.reshape(-1)]).to(torch.int32).contiguous()
    # the loss denominator (system.py:189,196) from the labels, on the host: no device round trip
    chars = sum(len(s) for s in labels)
    total = (chars + len(labels)) * min(K, 2) + chars * max(K - 2, 0)
    masks = torch.stack([generate_attn_masks(p)[1] for p in perms.cpu()]).to(torch.uint8).to(dev).contiguous()
    padding = ((tgt_in == system.pad_id) | (tgt_in == system.eos_id)).to(torch.uint8).contiguous()
    tokens = tgt_in.to(torch.int32).contiguous()
    if memory is None:
        memory = model.encode(images)
    memory = memory.float().contiguous()
    native = model._sync_native().model
    flat = torch.zeros(lib.parseq_model_grad_elems(native), dtype=torch.float32, device=dev)
    dmemory = torch.empty_like(memory)
    ws_bytes = lib.parseq_train_decoder_workspace_bytes(native, B, L, K)
    workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    loss = torch.empty(1 + K, dtype=torch.float32, device=dev)
    _native.check(lib.parseq_train_decoder(native, _native.ptr(memory), _native.ptr(tokens), _native.ptr(targets), _native.ptr(padding),
                                           _native.ptr(masks), B, L, K, total, _native.ptr(loss), _native.ptr(flat), _native.ptr(dmemory),
                                           _native.ptr(workspace), ws_bytes, _native.stream_ptr()))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    return DecoderBackward(loss=loss[0], perm_losses=loss[1:], grads=param_views(native, flat, shapes), flat=flat, dmemory=dmemory, perms=perms,
                           workspace=workspace, _shape=(B, L, K), _model=native)


def loss_and_grads(system, images: Tensor, labels, perms: Optional[Tensor] = None) -> DecoderBackward:
    """Loss and the gradient of EVERY parameter for one batch — the state `loss.backward()` leaves after the reference's
    `training_step` (system.py:168-199), dropout off.  `images`: fp32 [B, 3, H, W] on the device, normalised."""
    lib = _native.lib()
    model = system.model
    images = model._check_images(images)
    if images.dtype != torch.float32:
        images = ((images.float() / 255.0) - 0.5) / 0.5 if images.dtype == torch.uint8 else images.float()
    native = model._sync_native().model
    B = images.shape[0]
    ws_bytes = lib.parseq_train_encoder_workspace_bytes(native, B)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=images.device)
    memory = torch.empty(B, model.encoder.pos_embed.shape[1], model._cfg['embed_dim'], dtype=torch.float32, device=images.device)
    _native.check(lib.parseq_train_encoder_forward(native, _native.ptr(images), B, _native.ptr(memory), _native.ptr(ws), ws_bytes,
                                                   _native.stream_ptr()))
    res = decoder_backward(system, images, labels, perms, memory=memory)
    _native.check(lib.parseq_train_encoder_backward(native, _native.ptr(res.dmemory), B, _native.ptr(res.flat), _native.ptr(ws), ws_bytes,
                                                    _native.stream_ptr()))
    res.memory = memory
    return res
