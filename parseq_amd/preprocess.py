"""Input step on the device.

The reference's transform (strhub/data/module.py:69-82) is Resize(img_size, BICUBIC) -> ToTensor -> Normalize(0.5, 0.5) on
PIL images.  `resize_batch` is the first step (Pillow's 8-bit bicubic resampling, bit-exact) as a HIP kernel over a ragged
batch of uint8 HWC images that already live in device memory; its uint8 [N, 3, H, W] result goes straight into
`model(images)`, whose patch-embed loader applies ToTensor + Normalize (images_dtype = PARSEQ_U8).
"""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor

from . import _native


def resize_batch(images: Sequence[Tensor], size=(32, 128)) -> Tensor:
    """images: CUDA uint8 tensors [H_i, W_i, 3] (sizes may differ).  Returns uint8 [N, 3, size[0], size[1]]."""
    if len(images) == 0:
        raise ValueError('empty batch')
    dev = images[0].device
    descs = (_native.ImageDesc * len(images))()
    keep = []
    for i, im in enumerate(images):
        if not im.is_cuda:
            raise RuntimeError('resize_batch runs on the GPU (no CPU fallback); move the decoded images to the device first')
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError(f'image {i}: expected uint8 [H, W, 3], got {im.dtype} {list(im.shape)}')
        if im.stride(2) != 1 or im.stride(1) != 3:
            im = im.contiguous()
        keep.append(im)
        descs[i].data = im.data_ptr()
        descs[i].height, descs[i].width, descs[i].row_stride = im.shape[0], im.shape[1], im.stride(0)
    lib = _native.lib()
    out = torch.empty((len(images), 3, size[0], size[1]), dtype=torch.uint8, device=dev)
    ws = torch.empty((lib.parseq_resize_workspace_bytes(len(images)),), dtype=torch.uint8, device=dev)
    with _native.guard(dev):
        _native.check(lib.parseq_resize_bicubic(descs, len(images), size[0], size[1], _native.ptr(out), _native.ptr(ws), _native.stream_ptr(dev)))
    # the descriptor array is host memory read by an asynchronous copy: keep it (and the inputs) alive until the stream has passed
    torch.cuda.current_stream(dev).synchronize()
    return out
