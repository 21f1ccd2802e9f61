"""Shared helpers for the -m gpu parity tests (everything goes through the C ABI via parseq_amd)."""
import ctypes as C

import torch

from oracle.synth import CONFIGS, synth_state_dict

DEV = 'cuda'


def make_model(name, precision, decode_ar=True, refine_iters=1, seed=0):
    from parseq_amd import create_model
    m = create_model(name, decode_ar=decode_ar, refine_iters=refine_iters, precision=precision)
    m.model.load_state_dict(synth_state_dict(CONFIGS[name], seed))
    return m.eval().to(DEV)


def report(tag, got, want):
    d = (got.float().cpu() - want.float().cpu()).abs()
    idx = int(d.flatten().argmax())
    msg = (f'[{tag}] shape {tuple(got.shape)} max|d| {d.max().item():.3e} mean|d| {d.mean().item():.3e} '
           f'|want|max {want.abs().max().item():.3e} worst flat idx {idx} '
           f'nan_got {int(torch.isnan(got.float()).sum())}')
    print(msg)
    return d.max().item(), msg


def native():
    from parseq_amd import _native
    return _native, _native.lib()
