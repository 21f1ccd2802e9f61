"""How often does the bf16 throughput mode decode the same string as the fp32 parity mode?  Synthetic ('trained-like')
weights, 4096 seeded random crops, AR + 1 refinement.  Reports string agreement, per-position arg-max agreement and the
distribution of the fp32 top-1 / top-2 margin at the positions that differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle.synth import CONFIGS, synth_images, synth_state_dict
from parseq_amd import create_model


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'parseq'
    out = {}
    x = synth_images(4096, CONFIGS[name], seed=99)
    for prec in ('fp32', 'bf16'):
        m = create_model(name, precision=prec, refine_iters=int(os.environ.get('REFINE', '1')))
        m.model.load_state_dict(synth_state_dict(CONFIGS[name], 0))
        m = m.eval().to('cuda')
        with torch.inference_mode():
            lo = torch.cat([m(x[i:i + 512].to('cuda'), 25).float().cpu() for i in range(0, 4096, 512)])
        out[prec] = (lo, m.tokenizer.decode(lo.softmax(-1))[0])
    lf, sf = out['fp32']
    lb, sb = out['bf16']
    same_str = sum(a == b for a, b in zip(sf, sb))
    # positions that matter: up to and including the first EOS of the fp32 decode
    ids_f, ids_b = lf.argmax(-1), lb.argmax(-1)
    eos = (ids_f == 0).int().argmax(-1) + ((ids_f == 0).sum(-1) == 0) * 26
    mask = torch.arange(26)[None, :] <= eos[:, None]
    agree = ((ids_f == ids_b) & mask).sum().item(), mask.sum().item()
    top2 = lf.topk(2, -1).values
    margin = (top2[..., 0] - top2[..., 1])
    diff = (ids_f != ids_b) & mask
    print(f'{name}: strings identical {same_str}/4096 = {100 * same_str / 4096:.2f} %; positions (up to first EOS) identical '
          f'{agree[0]}/{agree[1]} = {100 * agree[0] / agree[1]:.3f} %')
    first = torch.where(diff.any(-1), diff.int().argmax(-1), torch.full((4096,), -1))
    rows = (first >= 0).nonzero().flatten()
    fm = margin[rows, first[rows]]
    dl = (lf - lb).abs().amax(-1)[rows, first[rows]]
    print(f'rows with a differing position: {len(rows)}; fp32 top-1/top-2 margin at the FIRST differing position: max {float(fm.max()):.3e}, '
          f'median {float(fm.median()):.3e}; |dlogit| there: max {float(dl.max()):.3e}')
    print(f'max |dlogit| {float((lf - lb).abs().max()):.3e}; fp32 margin at differing positions: '
          f'max {float(margin[diff].max()) if diff.any() else 0:.3e}, median {float(margin[diff].median()) if diff.any() else 0:.3e}; '
          f'median margin overall {float(margin[mask].median()):.3e}')


if __name__ == '__main__':
    main()
