"""CPU, world_size 2, gloo: the data-parallel wrapper (parseq_amd/parallel.py) — shard bounds, all-gather of logits
(equal and ragged shards), and the end-to-end sharded forward with a stand-in model (the HIP model needs a GPU; the
collective logic does not)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parseq_amd.parallel import all_gather_logits, data_parallel_forward, shard_bounds


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _FakeModel:
    """Deterministic per-image 'logits' so the gathered result can be checked against a single-process run."""

    def __call__(self, images, max_length=None):
        L = 26 if max_length is None else min(max_length, 25) + 1
        base = images.flatten(1).sum(1)                      # [b]
        return base[:, None, None] + torch.arange(L)[None, :, None] * 0.5 + torch.arange(95)[None, None, :] * 0.01


class _FakeEarlyExitModel(_FakeModel):
    """Adds the batch-level early exit of model.py:144-145: every image has its own 'first EOS step', the batch returns
    max over its rows + 1 positions — so shards of one batch can disagree on L, and `forward_with_length` reports it."""

    @staticmethod
    def _eos_step(images):
        return (images.flatten(1).sum(1) * 1000).long() % 20          # per image, in [0, 20)

    def forward_with_length(self, images, max_length=None):
        full = _FakeModel.__call__(self, images, max_length)
        if max_length is not None:
            return full, full.shape[1]
        return full, int(self._eos_step(images).max()) + 1

    def __call__(self, images, max_length=None):
        full, L = self.forward_with_length(images, max_length)
        return full[:, :L]


def _worker(rank, world, port, n_images, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        images = torch.rand(n_images, 3, 32, 128, generator=g)
        model = _FakeModel()
        full = data_parallel_forward(model, images, None)
        want = model(images)
        ok1 = torch.equal(full, want)
        full7 = data_parallel_forward(model, images, 7)
        ok2 = full7.shape == (n_images, 8, 95) and torch.equal(full7, model(images, 7))
        # mismatching L across ranks must be refused, not silently concatenated
        try:
            all_gather_logits(torch.zeros(2, 26 - rank, 95))
            ok3 = False
        except RuntimeError:
            ok3 = True
        # early exit under sharding (SURVEY 8e option ii): shards disagree on L, the wrapper agrees on the single-device length
        ee = _FakeEarlyExitModel()
        steps = ee._eos_step(images)
        lo, hi = shard_bounds(n_images, world, rank)
        assert len({int(steps[a:b].max()) for a, b in (shard_bounds(n_images, world, r) for r in range(world))}) > 1, 'test needs differing shard lengths'
        got = data_parallel_forward(ee, images, None)
        want_ee = ee(images)
        ok3 = ok3 and got.shape == want_ee.shape and torch.equal(got, want_ee) and got.shape[1] == int(steps.max()) + 1
        # uniform fast path (what bench.py's multi-GPU loop uses): one collective, rank order preserved
        mine = torch.full((3, 26, 95), float(rank))
        both = all_gather_logits(mine, uniform=True)
        ok4 = both.shape == (3 * world, 26, 95) and all(bool((both[3 * r:3 * r + 3] == r).all()) for r in range(world))
        q.put((rank, ok1, ok2, ok3 and ok4))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_images', [8, 9])      # equal shards, ragged shards
def test_two_rank_gloo_sharded_forward(n_images):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in results), results


def _one_rank_worker(port, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        x = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
        same = all_gather_logits(x, uniform=True)                       # world 1: returned as is
        forced = all_gather_logits(x, uniform=True, force=True)         # the collective really runs (bench.py --force-dist)
        q.put((same is x, bool(torch.equal(forced, x)), forced.data_ptr() != x.data_ptr()))
    finally:
        dist.destroy_process_group()


def test_one_rank_forced_collective():
    """bench.py --force-dist: with a one-rank group the gather is skipped unless forced; forced, it returns an equal copy."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pr = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    pr.start()
    got = q.get(timeout=120)
    pr.join(timeout=60)
    assert got == (True, True, True)


def _grad_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from parseq_amd.parallel import average_gradients
        n = 1000 + 37                                        # not a multiple of the bucket size: the last bucket is ragged
        mine = torch.arange(n, dtype=torch.float32) * (rank + 1)
        out = average_gradients(mine, bucket_elems=256)
        want = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        q.put((rank, out.data_ptr() == mine.data_ptr() and torch.allclose(out, want, rtol=1e-6, atol=0)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_average():
    """Row N3's data-parallel step: the flat gradient buffer is averaged across ranks in place, bucket by bucket."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1] and all(r[1] for r in results)


def _segment_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from parseq_amd.parallel import average_gradient_segments, average_gradients
        n = 5000
        mine = (torch.arange(n, dtype=torch.float32) % 97) * (rank + 1) + rank
        ref = average_gradients(mine.clone(), bucket_elems=512)
        # completion order of the training step's backward: the tail of the buffer first, then the middle ranges back to front, the head last
        segs = [(4000, 5000, None), (3000, 4000, None), (2000, 3000, None), (700, 2000, None), (0, 700, None)]
        out = average_gradient_segments(mine, segs)
        ok = out.data_ptr() == mine.data_ptr() and torch.equal(out, ref)
        try:
            average_gradient_segments(mine.clone(), [(0, 700, None), (800, 5000, None)])     # a hole: refused before any collective is issued
            ok = False
        except ValueError:
            pass
        try:
            average_gradient_segments(mine.clone(), [])                                      # nothing to reduce: refused, not an IndexError
            ok = False
        except ValueError:
            pass
        try:
            # a side stream without the events that order it behind the backward: the collective would race the gradient writers — refused
            average_gradient_segments(mine.clone(), segs, comm_stream=object())
            ok = False
        except ValueError:
            pass
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_segments_equal_the_bucketed_average():
    """The overlapped form of the data-parallel gradient step (parallel.average_gradient_segments: one all-reduce per gradient segment
    in the order the backward finishes them — reference train.py:65-71, DDP's reducer) gives bit for bit what the one-pass bucketed
    average gives, and refuses a segment list that does not tile the buffer."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_segment_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1] and all(r[1] for r in results)


# ---- the real model + RCCL on two GPUs (skipped on a 1-GPU box: the driver's multi-GPU tier and any 2+-GPU box run it) ----

def _nccl_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', rank)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)      # 'nccl' is RCCL on ROCm
    try:
        from oracle.synth import CONFIGS, synth_images, synth_state_dict
        from parseq_amd import create_model
        name = 'parseq'
        cfg = CONFIGS[name]
        sd = synth_state_dict(cfg, 0)
        images = synth_images(17, cfg, seed=7).to(dev)           # 17: ragged shards (9 + 8); with seed 7 the shards exit after 6 and 13
        # positions on their own (fp32 oracle), so the agreed length (13) really is a maximum over disagreeing shards
        results = []
        for refine, max_length in ((1, None), (0, None), (0, 25)):      # AR+1; AR+0 with the natural early exit; AR+0 forced
            m = create_model(name, decode_ar=True, refine_iters=refine, precision='bf16')
            m.model.load_state_dict(sd)
            m = m.eval().to(dev)
            with torch.inference_mode():
                sharded = data_parallel_forward(m, images, max_length)
                single = m(images, max_length)                   # this rank alone, whole batch
            results.append((tuple(sharded.shape) == tuple(single.shape), bool(torch.equal(sharded, single)), tuple(single.shape)))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_nccl_real_model_matches_single_rank_bit_for_bit():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs (the 1-GPU box cannot run RCCL between ranks)')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, res in results:
        assert all(shape_ok and equal for shape_ok, equal, _ in res), (rank, res)
    # the natural-exit run must really have exited early on these weights (synth.py lifts the EOS bias), or the test is vacuous
    assert results[0][1][1][2][1] < 26, results[0][1]


def _bench_line(args, env_extra, timeout):
    """Run bench.py as the driver does (a fresh interpreter) and return (the JSON lines it printed, the full stdout, stderr tail)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    lines = []
    for ln in r.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith('{') and ln.endswith('}'):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    return r.returncode, lines, r.stdout, r.stderr[-2000:]


def test_bench_py_two_ranks_print_one_line_for_the_whole_job():
    """bench.py's own multi-rank path on CPU (PARSEQ_BENCH_STUB=1: gloo, a shape-only stand-in for the model): `--gpus 2` from a plain
    shell re-executes under torch.distributed.run on 127.0.0.1, both ranks run the timed loop with the uniform all-gather inside it,
    and rank 0 alone prints ONE JSON line whose value / config describe the whole job (2 x 512 crops per step)."""
    rc, lines, out, err = _bench_line(['--gpus', '2', '--steps', '3', '--warmup', '1', '--repeats', '2'], {'PARSEQ_BENCH_STUB': '1'}, 600)
    assert rc == 0, err
    assert len(lines) == 1, out
    d = lines[0]
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['stub'] is True
    assert d['config']['global_batch'] == 1024 and d['config']['parallelism'].startswith('dp2')
    assert d['config']['output_shape'] == [1024, 26, 95]                 # every rank ends a step holding the logits of all 1024 crops
    assert abs(d['value'] - 1024 * 3 / (d['ms_per_step'] * 3e-3)) <= 1e-3 * d['value']      # whole-job crops / max-over-ranks time
    assert 'stub' in d['data']                                           # and the line cannot be mistaken for a measurement
    # the training leg's multi-rank form: every rank steps on its own 384-crop shard, rank 0 reports the whole job (configs[4]: 8 x 384)
    assert d['train']['stub'] is True and d['train']['n_gpus'] == 2 and d['train']['global_batch'] == 768 and d['train']['value'] > 0


def test_bench_py_refuses_a_world_size_that_contradicts_gpus():
    rc, lines, out, err = _bench_line(['--gpus', '1', '--steps', '1', '--warmup', '0'], {'PARSEQ_BENCH_STUB': '1', 'WORLD_SIZE': '2', 'RANK': '0'}, 300)
    assert rc != 0 and not lines


@pytest.mark.gpu
def test_bench_py_force_dist_runs_the_rccl_path_on_one_gpu():
    """The distributed path of bench.py on the one GPU a test box has: RCCL process group of one rank, the real all_gather_into_tensor
    inside the timed step, two batches in flight on separate streams next to the communicator's streams (GPU_MAX_HW_QUEUES=8, DESIGN.md
    section 6).  Short: 5 steps, no profile / CPU baseline / parity legs."""
    rc, lines, out, err = _bench_line(['--force-dist', '--steps', '5', '--warmup', '2', '--repeats', '2', '--no-profile', '--no-cpu-baseline', '--no-parity', '--no-train', '--no-natural-exit', '--no-throughput-mode'],
                                      {}, 900)
    assert rc == 0, err
    assert len(lines) == 1, out
    d = lines[0]
    assert d['n_gpus'] == 1 and d['config']['output_shape'] == [512, 26, 95] and d['value'] > 0 and d['sequential_value'] > 0
    # the same run without the process group: the collectives must not change a bit of the logits (two batches in flight in both)
    rc2, lines2, out2, err2 = _bench_line(['--steps', '5', '--warmup', '2', '--repeats', '2', '--no-profile', '--no-cpu-baseline', '--no-parity', '--no-train',
                                           '--no-natural-exit', '--no-throughput-mode'], {}, 900)
    assert rc2 == 0 and len(lines2) == 1, err2
    assert d['config']['steps_in_flight'] == lines2[0]['config']['steps_in_flight'] >= 2
    assert d['output_sha256_16'] == lines2[0]['output_sha256_16'], (d['output_sha256_16'], lines2[0]['output_sha256_16'])


def test_scale_curve_tool_under_the_stub(tmp_path):
    """tools/scale_curve.py — the one command that turns a multi-GPU box into profiles/scale.json — on CPU under PARSEQ_BENCH_STUB=1 at N = 1, 2: one row per N with the
    whole-job value, the global batch, the parallelism string and the weak-scaling efficiency against N = 1; the record says it is a stub."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / 'scale.json'
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'scale_curve.py'), '--gpus', '1', '2', '--out', str(out), '--stub', '--', '--steps', '3', '--warmup', '1',
                        '--repeats', '2'], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    rows = {row['n_gpus']: row for row in rec['rows']}
    assert set(rows) == {1, 2} and all('value' in row and row['stub'] for row in rows.values()), rec
    assert rows[1]['global_batch'] == 512 and rows[2]['global_batch'] == 1024 and rows[2]['parallelism'].startswith('dp2')
    assert rows[1]['weak_scaling_efficiency'] == 1.0 and rows[2]['weak_scaling_efficiency'] > 0
    assert rows[2]['train_value'] is not None                      # the training leg's multi-rank plumbing ran too
