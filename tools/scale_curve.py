#!/usr/bin/env python3
"""One command from a 1 -> 8 GPU curve: runs `bench.py --gpus N` for every N given (default 1 2 4 8, each as its own
process: bench.py spawns its ranks under torch.distributed.run on 127.0.0.1) and writes profiles/scale.json — per N the whole-job `value`, `sequential_value`,
`train.value`, ms per step, the GPU_MAX_HW_QUEUES in effect, and the weak-scaling efficiency against N = 1 (value_N / (N * value_1)).  An N the box cannot serve is
recorded with bench.py's refusal, not skipped silently.  `--stub`: the plumbing on CPU (PARSEQ_BENCH_STUB=1, gloo; tests/test_parallel.py drives it at N = 1, 2).

    python tools/scale_curve.py [--gpus 1 2 4 8] [--out profiles/scale.json] [--stub] [-- <extra bench.py arguments>]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if '--' in argv:
        k = argv.index('--')
        argv, extra = argv[:k], argv[k + 1:]
    gpus, out, stub = [1, 2, 4, 8], os.path.join(ROOT, 'profiles', 'scale.json'), False
    i = 0
    while i < len(argv):
        if argv[i] == '--gpus':
            gpus = []
            i += 1
            while i < len(argv) and not argv[i].startswith('--'):
                gpus.append(int(argv[i]))
                i += 1
            continue
        if argv[i] == '--out':
            out = argv[i + 1]
            i += 2
            continue
        if argv[i] == '--stub':
            stub = True
        i += 1
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    if stub:
        env['PARSEQ_BENCH_STUB'] = '1'
    rows = []
    for n in gpus:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n)] + extra, env=env, capture_output=True, text=True, cwd=ROOT)
        line = None
        for ln in r.stdout.splitlines():
            ln = ln.strip()
            if ln.startswith('{') and ln.endswith('}'):
                try:
                    line = json.loads(ln)
                except ValueError:
                    pass
        if r.returncode != 0 or line is None:
            rows.append({'n_gpus': n, 'error': (r.stderr.strip().splitlines() or ['no JSON line'])[-1][:300], 'returncode': r.returncode})
            continue
        tr = line.get('train') or {}
        rows.append({'n_gpus': n, 'value': line['value'], 'sequential_value': line.get('sequential_value'), 'ms_per_step': line['ms_per_step'],
                     'global_batch': line['config']['global_batch'], 'parallelism': line['config']['parallelism'], 'train_value': tr.get('value'),
                     'train_ms_per_step': tr.get('ms_per_step'), 'gpu_max_hw_queues': '8 (bench.py default with a communicator)' if n > 1 else os.environ.get('GPU_MAX_HW_QUEUES', 'runtime default'),
                     'stub': bool(line.get('stub'))})
    base = next((r for r in rows if r.get('n_gpus') == 1 and 'value' in r), None)
    for r in rows:
        if base and 'value' in r:
            r['weak_scaling_efficiency'] = round(r['value'] / (r['n_gpus'] * base['value']), 4)
            if r.get('train_value') and base.get('train_value'):
                r['train_weak_scaling_efficiency'] = round(r['train_value'] / (r['n_gpus'] * base['train_value']), 4)
    rec = {'metric': 'images/sec (32x128 crops) PARSeq-S AR+refine; whole job', 'scaling': 'weak (512 crops per GPU per step; training: 384 per GPU)', 'rows': rows,
           'note': 'efficiency = value_N / (N x value_1), computed here for convenience only: the driver computes its own from the per-N lines'}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
