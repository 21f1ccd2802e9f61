#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, kernel-trace only).

MfmaUtil (rocprofiler's gfx94x formula; ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md section rocprofv3 PMC slots):
    100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CU_NUM * 4)
SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs of all XCDs; GRBM_GUI_ACTIVE is reported per XCD (the value stored is the
sum over the 8 XCDs when rocprofv3 aggregates dimensions, so it is divided by the XCD count below if it exceeds the kernel's
duration in cycles by more than 4x).
Usage: python tools/pmc_mfma_summary.py <results.db> [--clock-ghz 2.4]"""
import re
import sqlite3
import sys

CUS, SIMDS, XCDS = 256, 4, 8


def short(name):
    name = re.sub(r'\(.*$', '', name).replace('pq::', '').replace('void ', '')
    return re.sub(r'__hip_bfloat16|__bf16|DF16b', 'bf16', name)[:100]


def main():
    db = sqlite3.connect(sys.argv[1])
    ghz = float(sys.argv[sys.argv.index('--clock-ghz') + 1]) if '--clock-ghz' in sys.argv else 2.4
    rows = db.execute('select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection '
                      'group by kernel_name, counter_name').fetchall()
    per = {}
    for k, c, n, v, dur in rows:
        per.setdefault(k, {})[c] = (n, v, dur)
    print('| kernel | dispatches | avg us | MFMA busy cycles (all SIMDs) | GUI active cycles | MFMA util % (busy / (active x 256 CU x 4 SIMD)) | util % from duration x clock |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for k, d in sorted(per.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0, 0))[0] * kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0, 0))[1]):
        if 'SQ_VALU_MFMA_BUSY_CYCLES' not in d:
            continue
        n, busy, dur = d['SQ_VALU_MFMA_BUSY_CYCLES']
        active = d.get('GRBM_GUI_ACTIVE', (0, 0.0, 0))[1]
        cyc = dur * ghz                                     # ns * GHz = cycles
        if active > 4 * cyc:
            active /= XCDS
        u1 = 100.0 * busy / (active * CUS * SIMDS) if active else float('nan')
        u2 = 100.0 * busy / (cyc * CUS * SIMDS) if cyc else float('nan')
        if busy < 1:
            continue
        print(f'| `{short(k)}` | {n} | {dur / 1e3:.1f} | {busy:.3e} | {active:.3e} | {u1:.1f} | {u2:.1f} |')


if __name__ == '__main__':
    main()
