#!/usr/bin/env python3
"""Per (kernel, grid) statistics of a rocprofv3 --kernel-trace results.db: which launch SHAPES of one kernel carry its time.
Usage: python tools/rocprof_by_grid.py <results.db> <kernel name pattern> [top N]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    gcols = [c for c in cols if c.lower() in ('grid_x', 'grid_y', 'grid_z', 'grid_size_x', 'grid_size_y', 'grid_size_z')]
    wcols = [c for c in cols if c.lower() in ('workgroup_x', 'workgroup_y', 'workgroup_z', 'workgroup_size_x', 'workgroup_size_y', 'workgroup_size_z')]
    if not gcols:
        print('no grid columns in', cols)
        return
    sel = ', '.join(gcols + wcols)
    rows = db.execute(f'select {sel}, count(*), sum(end - start), min(end - start), max(end - start) from kernels where {namecol} like ? group by {sel} order by 3 desc',
                      ('%' + pat + '%',)).fetchall()
    total = sum(r[-3] for r in rows) or 1
    print(f'| grid {" x ".join(gcols)} (work-items) | block | launches | total ms | avg us | min us | max us | share |')
    print('|---|---|---:|---:|---:|---:|---:|---:|')
    for r in rows[:top]:
        g, w = r[:len(gcols)], r[len(gcols):len(gcols) + len(wcols)]
        n, tot, mn, mx = r[-4:]
        print(f'| {" x ".join(str(x) for x in g)} | {" x ".join(str(x) for x in w)} | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f}% |')
    print(f'\n{pat}: {total / 1e6:.3f} ms over {sum(r[-4] for r in rows)} launches, {len(rows)} distinct shapes')


if __name__ == '__main__':
    main()
