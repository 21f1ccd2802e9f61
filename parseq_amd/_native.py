"""ctypes binding of libparseq_hip.so (C ABI: include/parseq_hip.h).

This is the only way compute reaches the GPU in this package.  If the library is missing or fails to load the import
of this module's `lib()` raises — there is deliberately no fallback (a silent eager/CPU path would void every parity
and performance claim).  Device memory, streams and the caching allocator are torch's: tensors are passed as raw
device pointers + the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

PARSEQ_F32, PARSEQ_BF16, PARSEQ_U8, PARSEQ_BF16X3 = 0, 1, 2, 3
ARCH_PARSEQ, ARCH_VITSTR = 0, 1
FLAG_DECODE_AR, FLAG_TESTING, FLAG_LATENCY = 1, 2, 4
ABI_VERSION = 8


class ParseqConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'img_h', 'img_w', 'patch_h', 'patch_w', 'embed_dim', 'enc_depth', 'enc_heads', 'enc_mlp_ratio',
        'dec_depth', 'dec_heads', 'dec_mlp_ratio', 'num_tokens', 'max_label_length', 'bos_id', 'eos_id', 'pad_id')]
    _fields_ += [('enc_ln_eps', C.c_float), ('dec_ln_eps', C.c_float), ('arch', C.c_int32)]


class ImageDesc(C.Structure):
    _fields_ = [('data', C.c_void_p), ('height', C.c_int32), ('width', C.c_int32), ('row_stride', C.c_int64)]


class NativeError(RuntimeError):
    """A libparseq_hip entry point returned a non-zero status."""


_LIB: Optional[C.CDLL] = None

# name: (restype, argtypes) — must list every symbol include/parseq_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    'parseq_abi_version': (C.c_int, []),
    'parseq_last_error': (C.c_char_p, []),
    'parseq_model_create': (C.c_int, [C.POINTER(ParseqConfig), C.POINTER(C.c_void_p)]),
    'parseq_model_destroy': (None, [C.c_void_p]),
    'parseq_model_set_param': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'parseq_model_num_params': (C.c_int, [C.c_void_p]),
    'parseq_model_param_info': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    'parseq_plan_create': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    'parseq_plan_refresh': (C.c_int, [C.c_void_p, C.c_void_p]),
    'parseq_plan_create_ex': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    'parseq_plan_destroy': (None, [C.c_void_p]),
    'parseq_plan_workspace_bytes': (C.c_size_t, [C.c_void_p]),
    'parseq_resize_workspace_bytes': (C.c_size_t, [C.c_int]),
    'parseq_resize_bicubic': (C.c_int, [C.POINTER(ImageDesc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_cross_entropy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_postprocess': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_plan_set_profiling': (C.c_int, [C.c_void_p, C.c_int]),
    'parseq_plan_get_profile': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'parseq_encode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'parseq_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.POINTER(C.c_int), C.c_void_p]),
    'parseq_decode_hidden': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_decode_query': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_set_memory': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'parseq_vitstr_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'parseq_decode_logits': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_model_param_offset': (C.c_int64, [C.c_void_p, C.c_int]),
    'parseq_model_grad_elems': (C.c_int64, [C.c_void_p]),
    'parseq_model_set_train_precision': (C.c_int, [C.c_void_p, C.c_int]),
    'parseq_train_decoder_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    'parseq_train_decoder_workspace_offset': (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    'parseq_train_decoder': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'parseq_train_encoder_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'parseq_train_encoder_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'parseq_train_encoder_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'parseq_train_grad_segments': (C.c_int, [C.c_void_p]),
    'parseq_train_grad_segment': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]),
    'parseq_stream_wait_event': (C.c_int, [C.c_void_p, C.c_void_p]),
    'parseq_shard_bounds': (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'parseq_grad_norm': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_adamw_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    'parseq_model_get_param': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'parseq_model_get_params': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    'parseq_op_layernorm': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_void_p]),
    'parseq_op_linear': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p]),
    'parseq_op_ln_linear': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_void_p]),
    'parseq_op_split_pack': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'parseq_op_linear_cfg': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    'parseq_op_ln_linear_pairs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_float, C.c_void_p]),
    'parseq_op_ln_linear_gelu': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]),
    'parseq_op_mlp': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'parseq_op_mlp_variant': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'parseq_op_attn_fused': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'parseq_op_enc_blocks': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'parseq_op_enc_head_tail': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'parseq_op_enc_blocks_x3': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_op_enc_blocks_x3w': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_void_p]),
    'parseq_op_encoder_attention': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p]),
}


def lib_path() -> str:
    return os.environ.get('PARSEQ_HIP_LIB', _build.LIB_PATH)


def lib() -> C.CDLL:
    """Load (once) and return the shared library.  Raises if it is not there: build it with
    `python -m parseq_amd.build` / `__graft_entry__.build()`."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f'libparseq_hip.so not found at {path}. The PARSeq HIP backend has no CPU/eager fallback; '
                f'build it first: python -m parseq_amd.build')
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError here = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        if handle.parseq_abi_version() != ABI_VERSION:
            raise RuntimeError(f'libparseq_hip ABI {handle.parseq_abi_version()} != binding ABI {ABI_VERSION}')
        _LIB = handle
    return _LIB


def check(status: int) -> None:
    if status != 0:
        msg = lib().parseq_last_error()
        raise NativeError(f'libparseq_hip error {status}: {msg.decode() if msg else "?"}')


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)
RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class TorchPlanAllocator:
    """parseq_plan_create_ex's (alloc, release) pair on top of torch's caching allocator: a plan's arena is one uint8 tensor, alive until the library
    hands the block back (reference: every intermediate of strhub/models/parseq/model.py:86-169 is a caching-allocator block).  One instance serves any
    number of plans; `blocks` = {pointer: tensor} is what is out.  release() synchronises the device first: the caching allocator may hand the block
    to anyone the moment the tensor dies, and a plan's work may still be running on side streams (slots)."""

    def __init__(self, device):
        import torch
        self.device = torch.device(device)
        self.blocks = {}
        self.bytes_out = 0
        self.calls = 0
        self._alloc = ALLOC_FN(self._do_alloc)
        self._release = RELEASE_FN(self._do_release)

    def _do_alloc(self, nbytes, _user):
        import torch
        try:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        except Exception:
            return None
        self.blocks[t.data_ptr()] = t
        self.bytes_out += int(nbytes)
        self.calls += 1
        return t.data_ptr()

    def _do_release(self, ptr, _user):
        import torch
        t = self.blocks.pop(ptr, None)
        if t is not None:
            torch.cuda.synchronize(self.device)
            self.bytes_out -= t.numel()

    @property
    def alloc_ptr(self):
        return C.cast(self._alloc, C.c_void_p)

    @property
    def release_ptr(self):
        return C.cast(self._release, C.c_void_p)


def stream_ptr(device=None) -> C.c_void_p:
    """The current HIP stream of `device` (a torch.device, a tensor, or None = the current device).  Entry points that take
    a model or a plan run on that object's device whatever device is current (the library switches and restores), so the
    stream handed over must be that device's stream — pass the model's device / the tensors' device, never rely on the
    thread's current device."""
    import torch
    if device is not None and hasattr(device, 'device'):
        device = device.device
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def guard(device):
    """Context manager making `device` (torch.device or tensor) current: for the raw-pointer entry points (parseq_op_*,
    parseq_postprocess, parseq_resize_bicubic, parseq_cross_entropy, parseq_grad_norm), which launch on the current device."""
    import torch
    if hasattr(device, 'device'):
        device = device.device
    return torch.cuda.device(device)


def dtype_code(t) -> int:
    import torch
    if t == torch.float32:
        return PARSEQ_F32
    if t == torch.bfloat16:
        return PARSEQ_BF16
    if t == torch.uint8:
        return PARSEQ_U8
    raise TypeError(f'unsupported dtype {t}: libparseq_hip takes float32, bfloat16 or (images only) uint8')


def ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
