#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results.db (rocpd sqlite) into a per-kernel table: calls, total, avg, share.
Usage: python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--md] [--grid]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('pq::', '').replace('void ', '')
    name = re.sub(r'__hip_bfloat16|__bf16|DF16b', 'bf16', name)
    return name[:150]


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f'select {namecol}, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by {namecol} order by 3 desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f'| kernel | calls | total ms | avg us | min us | max us | share |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for name, n, tot, mn, mx in rows:
        print(f'| `{short(name)}` | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f}% |')
    print(f'\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')


if __name__ == '__main__':
    main()
