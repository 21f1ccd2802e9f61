#!/usr/bin/env python3
"""Timeline of the in-flight forward from a rocprofv3 --kernel-trace results.db: when each encoder launch runs, and how long the AR loop +
refinement of the same queue take behind it (wall clock), i.e. how the batches in flight interleave on the device.
Usage: python tools/timeline.py gpurun_out/prof/x_results.db [first_ms last_ms]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    qcol = next((c for c in cols if 'queue' in c), None) or next((c for c in cols if 'stream' in c), None)
    print('columns', cols, 'queue column', qcol)
    rows = db.execute(f'select {namecol}, start, end, {qcol} from kernels order by start').fetchall()
    t0 = rows[0][1]
    enc = [(s, e, q) for n, s, e, q in rows if 'enc_blocks' in n]
    print(f'{len(rows)} dispatches, {len(enc)} encoder launches, span {(rows[-1][2] - t0) / 1e6:.1f} ms')
    # per encoder launch: the decoder kernels of the same queue until that queue's next encoder launch
    out = []
    for i, (s, e, q) in enumerate(enc):
        nxt = next((s2 for s2, _, q2 in enc[i + 1:] if q2 == q), None)
        dec = [(n, ks, ke) for n, ks, ke, kq in rows if kq == q and ks >= e and (nxt is None or ks < nxt)]
        if not dec:
            continue
        d0, d1 = dec[0][1], max(ke for _, _, ke in dec)
        busy = sum(ke - ks for _, ks, ke in dec)
        # encoders of OTHER queues that overlap this decode window
        other = sum(max(0, min(e2, d1) - max(s2, d0)) for s2, e2, q2 in enc if q2 != q)
        out.append((s, e, q, d0, d1, len(dec), busy, other))
    print('| queue | encoder start ms | encoder ms | gap to first decoder kernel ms | decode window ms (first start .. last end) | kernels | their summed durations ms | other queues\' encoders inside the window ms |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    for s, e, q, d0, d1, n, busy, other in out[-24:]:
        print(f'| {q} | {(s - t0) / 1e6:.2f} | {(e - s) / 1e6:.2f} | {(d0 - e) / 1e6:.2f} | {(d1 - d0) / 1e6:.2f} | {n} | {busy / 1e6:.2f} | {other / 1e6:.2f} |')
    # overall: fraction of the span in which some encoder runs
    tail = enc[len(enc) // 2:]
    lo, hi = tail[0][0], tail[-1][1]
    cover = 0
    cur_s, cur_e = None, None
    for s, e, _ in tail:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += cur_e - cur_s
    print(f'\nsecond half of the run: {len(tail)} encoder launches over {(hi - lo) / 1e6:.2f} ms = {(hi - lo) / 1e6 / max(1, len(tail) - 1):.2f} ms per batch; some encoder running {100 * cover / (hi - lo):.1f} % of that time; '
          f'mean encoder duration {sum(e - s for s, e, _ in tail) / len(tail) / 1e6:.2f} ms')


if __name__ == '__main__':
    main()
