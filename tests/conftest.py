import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu on the GPU box)')


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should skip rather than crash at import of device memory.
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import json
    from safetensors.torch import load_file

    def _load(name):
        t = load_file(os.path.join(GOLDEN_DIR, f'{name}.safetensors'))
        with open(os.path.join(GOLDEN_DIR, f'{name}.json')) as f:
            meta = json.load(f)
        return t, meta
    return _load
