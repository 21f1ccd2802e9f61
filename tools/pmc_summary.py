#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes).  Values are KiB per dispatch; on gfx950 FETCH_SIZE counts 128-byte requests of wide coalesced reads at 64 B
(MI355X_MICROARCH.md section HBM), so the fetch side is reported raw and doubled.
Usage: python tools/pmc_summary.py <fetch_results.db> <write_results.db> [--json out.json] [--batch B]"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name).replace('pq::', '').replace('void ', '')
    return re.sub(r'__hip_bfloat16|__bf16|DF16b', 'bf16', name)[:110]


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute('select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name = ? group by kernel_name', (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


FAMILY = {'enc_blocks_kernel': 'enc.blocks_fused', 'enc_blocks_x3w_kernel': 'enc.blocks_x3', 'enc_blocks_x3_kernel': 'enc.blocks_x3', 'fused_attn_kernel': 'enc.attn_fused', 'fused_mlp_kernel': 'enc.mlp_fused', 'ln_panel_gemm_kernelILi384ENS_10PanelHeads': 'enc.qkv_gemm', 'attn_mfma_kernel': 'enc.attention',
          'dec_cross_attn_ar_kernel': 'dec.cross_attention'}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 512
    print('| kernel | dispatches | FETCH_SIZE MiB (raw / x2) | WRITE_SIZE MiB | avg us (profiled) |')
    print('|---|---:|---:|---:|---:|')
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        n, f_kib, dur = fetch[k]
        w_kib = write.get(k, (0, 0.0, 0))[1]
        print(f'| `{short(k)}` | {n} | {f_kib / 1024:.1f} / {2 * f_kib / 1024:.1f} | {w_kib / 1024:.1f} | {dur / 1e3:.1f} |')
        for pat, fam in FAMILY.items():
            if pat in k and fam not in out:
                out[fam] = {'fetch_bytes_raw': f_kib * 1024, 'fetch_bytes_x2_gfx950_correction': 2 * f_kib * 1024, 'write_bytes': w_kib * 1024,
                            'hbm_bytes': 2 * f_kib * 1024 + w_kib * 1024, 'dispatches': n, 'batch': batch}
    if '--json' in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
