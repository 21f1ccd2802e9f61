"""CPU tests of the drop-in boundary: the C-ABI library loads here (no GPU needed for dlopen) and exports every symbol
include/parseq_hip.h declares; calls that would compute fail loudly instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    from parseq_amd import build
    return build.build(verbose=False)


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'parseq_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(parseq_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound(built_lib):
    from parseq_amd import _native
    declared = _declared_symbols()
    assert len(declared) >= 15
    handle = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(handle, name), f'{name} declared in include/parseq_hip.h but not exported'
    assert sorted(_native.SIGNATURES) == declared, 'ctypes binding and header disagree'


def test_library_loads_and_reports_abi(built_lib):
    from parseq_amd import _native
    lib = _native.lib()
    assert lib.parseq_abi_version() == _native.ABI_VERSION


def test_no_device_means_loud_failure(built_lib):
    """Without a gfx950 device the library must refuse, not emulate."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from parseq_amd import _native
    lib = _native.lib()
    cfg = _native.ParseqConfig(img_h=32, img_w=128, patch_h=4, patch_w=8, embed_dim=384, enc_depth=12, enc_heads=6,
                               enc_mlp_ratio=4, dec_depth=1, dec_heads=12, dec_mlp_ratio=4, num_tokens=97,
                               max_label_length=25, bos_id=95, eos_id=0, pad_id=96, enc_ln_eps=1e-6, dec_ln_eps=1e-5)
    handle = ctypes.c_void_p(0)
    status = lib.parseq_model_create(ctypes.byref(cfg), ctypes.byref(handle))
    assert status != 0 and not handle.value
    assert lib.parseq_last_error()


def test_missing_library_raises(monkeypatch, tmp_path):
    from parseq_amd import _native
    monkeypatch.setattr(_native, '_LIB', None)
    monkeypatch.setenv('PARSEQ_HIP_LIB', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU/eager fallback'):
        _native.lib()
