#!/usr/bin/env python3
"""Average value per dispatch of every counter in one or more rocprofv3 --pmc results.db files, for kernels matching a pattern.
Usage: python tools/pmc_generic.py <pattern> <results.db> [<results.db> ...]"""
import sqlite3
import sys


def main():
    pat = sys.argv[1]
    print('| counter | dispatches | avg value | avg us |')
    print('|---|---:|---:|---:|')
    for path in sys.argv[2:]:
        db = sqlite3.connect(path)
        rows = db.execute('select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection '
                          'group by kernel_name, counter_name').fetchall()
        for k, c, n, v, dur in rows:
            if pat in k:
                print(f'| {c} | {n} | {v:.4e} | {dur / 1e3:.1f} |')


if __name__ == '__main__':
    main()
