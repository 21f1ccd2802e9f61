#!/usr/bin/env python3
"""Read text from image files — the command line of the reference's `read.py:27-48` on the MI355X backend.

    ./read.py pretrained=parseq --images a.png b.jpg [--device cuda] [name:type=value ...]
    ./read.py path/to/lightning.ckpt --images *.png refine_iters:int=2 decode_ar:bool=false

Differences from the reference, all on the device side of the same results: the files are decoded on the host (PIL),
everything after that — the bicubic resize of the reference transform (bit-exact with Pillow), ToTensor + Normalize, the
model, soft-max / greedy pick / EOS cut — runs in libparseq_hip, and all images go through ONE batched forward instead of a
Python loop of batch-1 calls.
"""
import argparse

import torch
from PIL import Image

from parseq_amd import load_from_checkpoint, parse_model_args
from parseq_amd.preprocess import resize_batch


@torch.inference_mode()
def read_files(model, files, device='cuda'):
    """[(file, label, confidence)] for image files, one batched forward."""
    import numpy as np
    crops = [torch.from_numpy(np.asarray(Image.open(f).convert('RGB')).copy()).to(device) for f in files]
    if not crops:
        return []
    batch = resize_batch(crops, tuple(model.hparams.img_size))          # uint8 [N, 3, H, W], Pillow-exact bicubic
    labels, confidences = model.tokenizer.read(model(batch))             # normalisation happens inside the patch embedding
    return list(zip(files, labels, confidences.tolist()))


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('checkpoint', help="Model checkpoint (or 'pretrained=<model_id>')")
    parser.add_argument('--images', nargs='+', help='Images to read')
    parser.add_argument('--device', default='cuda')
    args, unknown = parser.parse_known_args(argv)
    kwargs = parse_model_args(unknown)
    print(f'Additional keyword arguments: {kwargs}')

    model = load_from_checkpoint(args.checkpoint, **kwargs).eval().to(args.device)
    for fname, pred, _ in read_files(model, args.images or [], args.device):
        print(f'{fname}: {pred}')


if __name__ == '__main__':
    main()
