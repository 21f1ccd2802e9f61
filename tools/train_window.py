#!/usr/bin/env python3
"""Kernels of ONE phase of the training step from a rocprofv3 --kernel-trace results.db: the window between the last launch of
`--after <name fragment>` and the first launch of `--before <name fragment>` behind it (default: the decoder's window, between the
encoder's one-launch forward and the first kernel of the encoder's backward), of the LAST step in the trace.  Per kernel name: launches,
total and average duration; the window's wall time and the sum of its kernel durations (their difference = gaps between kernels).
Usage: python tools/train_window.py results.db [--after enc_blocks_kernel] [--before ln_bwd_kernel] [--list]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name).replace('pq::', '').replace('void ', '')
    return name[:110]


def main():
    args = sys.argv[1:]
    db = sqlite3.connect(args[0])
    opt = lambda k, d: args[args.index(k) + 1] if k in args else d
    after, before = opt('--after', 'enc_blocks_kernel'), opt('--before', 'train_attn_bf16_kernel')
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f'select {namecol}, start, end from kernels order by start').fetchall()
    starts = [i for i, r in enumerate(rows) if after in r[0]]
    if not starts:
        raise SystemExit(f'no kernel matching {after}')
    if after == before and len(starts) >= 2:      # one whole step: from the second-last launch of the kernel to the last
        i0, i1 = starts[-2], starts[-1]
    else:
        i0 = starts[-1]
        i1 = next((i for i in range(i0 + 1, len(rows)) if before in rows[i][0]), len(rows))
    win = rows[i0 + 1:i1]
    wall = (win[-1][2] - rows[i0][2]) / 1e3 if win else 0.0
    agg = {}
    for n, s, e in win:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1; a[1] += e - s
    tot = sum(v[1] for v in agg.values()) / 1e3
    print(f'window: {len(win)} launches between the end of `{after}` and the first `{before}`: wall {wall:.1f} us, sum of kernel durations {tot:.1f} us')
    print('| kernel | launches | total us | avg us |')
    print('|---|---:|---:|---:|')
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'| `{k}` | {n} | {t / 1e3:.1f} | {t / 1e3 / n:.1f} |')
    if '--list' in args:
        prev = rows[i0][2]
        for n, s, e in win:
            print(f'{(s - prev) / 1e3:8.1f} gap {(e - s) / 1e3:8.1f} us  {short(n)}')
            prev = e


if __name__ == '__main__':
    main()
