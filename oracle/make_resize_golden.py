"""Mint the resize fixtures (row N2): Pillow's own `Image.resize((128, 32), BICUBIC)` on seeded inputs.

Run in the build container (Pillow is installed here; it is the third-party routine torchvision's `T.Resize` calls for PIL
images, strhub/data/module.py:77).  Inputs are regenerated from the seed by the tests; only Pillow's outputs are stored.
    python oracle/make_resize_golden.py   ->  tests/golden/resize_pillow.npz
"""
import os

import numpy as np
import PIL
from PIL import Image

SIZES = [(32, 128), (31, 100), (64, 256), (17, 53), (200, 37), (48, 160), (100, 400), (33, 129), (5, 7), (300, 1200), (1, 1), (2, 300)]
OUT_H, OUT_W = 32, 128


def make_input(h: int, w: int, kind: int) -> np.ndarray:
    if kind == 0:
        return np.random.default_rng(1000 * h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([(xx * 3 + yy * 5) % 256, (xx * yy) % 256, (255 - xx) % 256], -1).astype(np.uint8)


def main():
    out = {}
    for h, w in SIZES:
        for kind in (0, 1):
            img = make_input(h, w, kind)
            out[f'{h}x{w}_{kind}'] = np.asarray(Image.fromarray(img, 'RGB').resize((OUT_W, OUT_H), Image.BICUBIC))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'resize_pillow.npz')
    np.savez_compressed(path, pillow_version=np.array(PIL.__version__), **out)
    print(path, len(out), 'cases, Pillow', PIL.__version__)


if __name__ == '__main__':
    main()
