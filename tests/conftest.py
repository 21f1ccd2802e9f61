import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'small_batch_route: runs with the library\'s default small-batch route (see _one_launch_encoder_at_every_batch)')


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


@pytest.fixture(autouse=True)
def _one_launch_encoder_at_every_batch(request, monkeypatch):
    """Since round 6 the bf16x3 encoder of PARSeq-S runs as per-operation launches up to batch 64 (lib_internal.h small_batch_max) and as
    the one-launch kernel above it.  The golden vectors are batches of 2-8 crops: without this fixture they would stop reaching the
    one-launch kernels they were minted to pin.  So every test pins the route off (PARSEQ_SMALL_BATCH=0, read when a plan is created) —
    except the ones marked `small_batch_route` (tests/test_small_batch.py), which hold the DEFAULT behaviour to the same goldens."""
    if 'small_batch_route' not in request.keywords:
        monkeypatch.setenv('PARSEQ_SMALL_BATCH', '0')
