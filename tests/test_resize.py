"""Row N2: bicubic resize of the input transform.

Fixtures: Pillow's own `Image.resize((128, 32), BICUBIC)` outputs on seeded inputs (oracle/make_resize_golden.py ->
tests/golden/resize_pillow.npz; Pillow is what torchvision's T.Resize calls for PIL images, strhub/data/module.py:77).
CPU: oracle/resize_oracle.py == Pillow bit for bit (stored outputs; live Pillow too when it is importable).
GPU: parseq_resize_bicubic (through the C ABI, ragged batch in one launch) == the same outputs bit for bit.
"""
import os

import numpy as np
import pytest
import torch

from oracle.make_resize_golden import OUT_H, OUT_W, SIZES, make_input
from oracle.resize_oracle import coefficients, resize_bicubic_u8

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'resize_pillow.npz'))
CASES = [(h, w, k) for h, w in SIZES for k in (0, 1)]


@pytest.mark.parametrize('h,w,kind', CASES)
def test_oracle_matches_pillow_fixtures(h, w, kind):
    got = resize_bicubic_u8(make_input(h, w, kind), OUT_H, OUT_W)
    assert np.array_equal(got, GOLD[f'{h}x{w}_{kind}'])


def test_oracle_matches_live_pillow_on_more_sizes():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(5)
    for _ in range(12):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 400))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, 'RGB').resize((OUT_W, OUT_H), Image.BICUBIC))
        assert np.array_equal(resize_bicubic_u8(img, OUT_H, OUT_W), ref), (h, w)


def test_coefficients_are_normalised():
    for n in (7, 128, 400, 1200):
        bounds, kk = coefficients(n, 128)
        sums = np.array([kk[i, :bounds[i, 1]].sum() for i in range(128)])
        assert np.all(np.abs(sums - (1 << 22)) <= bounds[:, 1])          # one rounding per tap


def test_resize_refuses_cpu():
    from parseq_amd.preprocess import resize_batch
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        resize_batch([torch.zeros(8, 8, 3, dtype=torch.uint8)])


@pytest.mark.gpu
def test_resize_kernel_matches_pillow_ragged_batch():
    from parseq_amd.preprocess import resize_batch
    imgs = [torch.from_numpy(make_input(h, w, k)).cuda() for h, w, k in CASES]
    out = resize_batch(imgs, (OUT_H, OUT_W)).cpu().numpy()              # [N, 3, 32, 128], one launch for all sizes
    for i, (h, w, k) in enumerate(CASES):
        want = GOLD[f'{h}x{w}_{k}'].transpose(2, 0, 1)
        assert np.array_equal(out[i], want), (h, w, k, int(np.abs(out[i].astype(int) - want.astype(int)).max()))


@pytest.mark.gpu
def test_resize_kernel_strided_rows_and_other_target():
    from parseq_amd.preprocess import resize_batch
    big = torch.from_numpy(make_input(64, 300, 0)).cuda()
    view = big[8:50, 20:260]                                             # non-contiguous rows (row stride 900 bytes)
    got = resize_batch([view], (16, 64)).cpu().numpy()[0].transpose(1, 2, 0)
    assert np.array_equal(got, resize_bicubic_u8(view.cpu().numpy().copy(), 16, 64))


@pytest.mark.gpu
def test_read_pipeline_uint8_images_to_strings():
    """read.py end to end on the device: ragged uint8 HWC crops -> resize -> (normalise in the patch embed) -> logits -> labels."""
    from gpu_util import DEV, make_model
    from oracle import parseq_oracle as O
    from parseq_amd.preprocess import resize_batch
    m = make_model('parseq', 'fp32')
    crops = [make_input(h, w, 1) for h, w in [(40, 150), (32, 128), (64, 200), (25, 90)]]
    x_dev = resize_batch([torch.from_numpy(c).to(DEV) for c in crops])
    x_ref = torch.stack([torch.from_numpy(resize_bicubic_u8(c, 32, 128)).permute(2, 0, 1) for c in crops])
    assert torch.equal(x_dev.cpu(), x_ref)
    with torch.inference_mode():
        got = m(x_dev, 25).float().cpu()
        want = m(O.normalize_u8(x_ref).to(DEV), 25).float().cpu()
    assert torch.equal(got, want)
    labels, conf = m.tokenizer.read(got.to(DEV))
    assert len(labels) == 4 and conf.shape == (4,)


@pytest.mark.gpu
def test_read_files_matches_reference_transform_pipeline(tmp_path):
    """read.py's read_files (files -> device resize -> uint8 forward -> device post-processing) prints the strings the reference pipeline
    (PIL resize -> ToTensor -> Normalize -> model -> tokenizer.decode) gives with the same weights (fp32 mode)."""
    Image = pytest.importorskip('PIL.Image')
    import read as read_cli
    from gpu_util import DEV, make_model
    from oracle import parseq_oracle as O
    files, tensors = [], []
    for i, (h, w) in enumerate([(40, 150), (32, 128), (64, 200)]):
        img = make_input(h, w, 1 - i % 2)
        f = tmp_path / f'crop{i}.png'
        Image.fromarray(img, 'RGB').save(f)
        files.append(str(f))
        ref = np.asarray(Image.fromarray(img, 'RGB').resize((128, 32), Image.BICUBIC))
        tensors.append(torch.from_numpy(ref.copy()).permute(2, 0, 1))
    m = make_model('parseq', 'fp32')
    out = read_cli.read_files(m, files, DEV)
    with torch.inference_mode():
        logits = m(O.normalize_u8(torch.stack(tensors)).to(DEV))
    want, probs = m.tokenizer.decode(logits.softmax(-1))
    assert [o[1] for o in out] == want
    assert all(abs(o[2] - float(p.prod())) <= 1e-4 * max(float(p.prod()), 1e-6) for o, p in zip(out, probs))
