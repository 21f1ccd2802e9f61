"""CPU tests of the drop-in boundary: the C-ABI library loads here (no GPU needed for dlopen) and exports every symbol
include/parseq_hip.h declares; calls that would compute fail loudly instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    from parseq_amd import build
    return build.build(verbose=False)


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'parseq_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(parseq_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound(built_lib):
    from parseq_amd import _native
    declared = _declared_symbols()
    assert len(declared) >= 15
    handle = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(handle, name), f'{name} declared in include/parseq_hip.h but not exported'
    assert sorted(_native.SIGNATURES) == declared, 'ctypes binding and header disagree'


def test_library_loads_and_reports_abi(built_lib):
    from parseq_amd import _native
    lib = _native.lib()
    assert lib.parseq_abi_version() == _native.ABI_VERSION


def test_shard_bounds_of_the_library_equal_the_python_mirror(built_lib):
    """parseq_shard_bounds (ABI 8): the split a caller without torch shards its crops with is the one parseq_amd.parallel uses — host
    arithmetic only, so it runs without a device — and nonsense arguments are refused."""
    from parseq_amd import _native
    from parseq_amd.parallel import shard_bounds
    lib = _native.lib()
    b, e = ctypes.c_int64(), ctypes.c_int64()
    for n in (0, 1, 7, 8, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert lib.parseq_shard_bounds(n, world, rank, ctypes.byref(b), ctypes.byref(e)) == 0
                assert (b.value, e.value) == shard_bounds(n, world, rank)
    for n, world, rank in ((-1, 2, 0), (8, 0, 0), (8, 2, 2), (8, 2, -1)):
        assert lib.parseq_shard_bounds(n, world, rank, ctypes.byref(b), ctypes.byref(e)) != 0
    assert lib.parseq_shard_bounds(8, 2, 0, None, ctypes.byref(e)) != 0


def test_no_device_means_loud_failure(built_lib):
    """Without a gfx950 device the library must refuse, not emulate."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from parseq_amd import _native
    lib = _native.lib()
    cfg = _native.ParseqConfig(img_h=32, img_w=128, patch_h=4, patch_w=8, embed_dim=384, enc_depth=12, enc_heads=6,
                               enc_mlp_ratio=4, dec_depth=1, dec_heads=12, dec_mlp_ratio=4, num_tokens=97,
                               max_label_length=25, bos_id=95, eos_id=0, pad_id=96, enc_ln_eps=1e-6, dec_ln_eps=1e-5)
    handle = ctypes.c_void_p(0)
    status = lib.parseq_model_create(ctypes.byref(cfg), ctypes.byref(handle))
    assert status != 0 and not handle.value
    assert lib.parseq_last_error()


def test_missing_library_raises(monkeypatch, tmp_path):
    from parseq_amd import _native
    monkeypatch.setattr(_native, '_LIB', None)
    monkeypatch.setenv('PARSEQ_HIP_LIB', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU/eager fallback'):
        _native.lib()


def _tool(*names):
    """First of `names` found on PATH or under the ROCm LLVM bin directory, else None."""
    import shutil
    for n in names:
        for cand in (shutil.which(n), os.path.join('/opt/rocm/lib/llvm/bin', n), os.path.join('/opt/rocm/llvm/bin', n)):
            if cand and os.path.exists(cand):
                return cand
    return None


def _gfx950_code_objects(lib_path, tmp_path):
    """The library is several translation units, each with its own offload bundle: every gfx950 code object, written to files."""
    import struct
    blob = open(lib_path, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    paths, at = [], blob.find(magic)
    while at >= 0:
        n = struct.unpack_from('<Q', blob, at + len(magic))[0]
        off = at + len(magic) + 8
        for _ in range(n):
            o, size, tlen = struct.unpack_from('<QQQ', blob, off)
            triple = blob[off + 24:off + 24 + tlen].decode()
            off += 24 + tlen
            if 'gfx950' in triple and size:
                path = tmp_path / f'lib{len(paths)}.co'
                path.write_bytes(blob[at + o:at + o + size])
                paths.append(path)
        at = blob.find(magic, at + len(magic))
    assert paths, 'no gfx950 code object in the library'
    return paths


def test_split_layernorm_loader_is_compiled_without_packed_f32(built_lib, tmp_path):
    """gemm.h ln_apply4: the bf16x3 GEMMs with the LayerNorm-fused A-loader produced wrong values (two workgroups per CU) while the
    loader's (x - mean) * rstd * gamma + beta was compiled to v_pk_mul_f32 / v_pk_fma_f32; empty-asm pins keep the SLP vectoriser
    from forming them.  A compiler bump could undo that silently, so the shipped code object is checked: the
    gemm_kernel<float, ..., ALayerNorm<float, E>, EpiStore / EpiGelu, SPLIT> instances (decoder q-projection / linear1 / head forms,
    whose epilogues have no packed arithmetic of their own) must not contain a packed-f32 multiply / fma / add."""
    import subprocess
    objdump, readelf, cxxfilt = _tool('llvm-objdump'), _tool('llvm-readelf'), _tool('c++filt', 'llvm-cxxfilt')
    if not (objdump and readelf and cxxfilt):
        pytest.skip('llvm-objdump / llvm-readelf / c++filt not found')
    want = []
    for path in _gfx950_code_objects(built_lib, tmp_path):
        syms = subprocess.run([readelf, '--symbols', '--wide', str(path)], capture_output=True, text=True, check=True).stdout
        mangled = sorted({ln.split()[-1] for ln in syms.splitlines() if ' FUNC ' in ln and ln.split()[-1].startswith('_Z')})
        names = subprocess.run([cxxfilt], input='\n'.join(mangled), capture_output=True, text=True, check=True).stdout.splitlines()
        want += [(path, m) for m, d in zip(mangled, names)
                 if 'gemm_kernel<float' in d and 'ALayerNorm<float' in d and ('EpiStore<float>' in d or 'EpiGelu<float>' in d)
                 and re.search(r'>, true, (false|true)>\(', d)]
    assert len(want) >= 6, f'expected the SPLIT ALayerNorm GEMM instances in the code object, found {len(want)}'
    for path, sym in want:
        asm = subprocess.run([objdump, '-d', '--no-show-raw-insn', f'--disassemble-symbols={sym}', str(path)],
                             capture_output=True, text=True, check=True).stdout
        assert 'v_mfma' in asm, f'{sym}: disassembly is empty?'
        bad = re.findall(r'v_pk_(?:mul|fma|add)_f32', asm)
        assert not bad, f'{sym}: {len(bad)} packed-f32 VALU instructions — the ln_apply4 pins no longer hold'


def test_occupancy_budgets_of_the_training_kernels(built_lib, tmp_path):
    """Round 3's training-step gains that are pure residency: the whole-tile all-bf16 GEMMs hold FOUR workgroups per CU only at <= 128 VGPRs
    without scratch (buffer loads with one VGPR offset per operand; `amdgpu_waves_per_eu(4, 4)` alone spilled 60 registers inside the loop),
    and the 32-key instantiation of the decoder attention kernel holds its many small workgroups only while it stays well under the 128-key
    one.  The numbers live in the shipped code object's metadata, so a compiler bump that silently undoes them fails here, on the CPU."""
    import subprocess
    readelf, cxxfilt = _tool('llvm-readelf'), _tool('c++filt', 'llvm-cxxfilt')
    if not (readelf and cxxfilt):
        pytest.skip('llvm-readelf / c++filt not found')
    meta = {}
    for path in _gfx950_code_objects(built_lib, tmp_path):
        notes = subprocess.run([readelf, '--notes', str(path)], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r'\n\s+- ', notes):
            name = re.search(r'\.name:\s+(\S+)', blk)
            if not name or '.vgpr_count' not in blk:
                continue
            get = lambda k: int((re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, -1])[1])      # noqa: E731   (-1: the note does not carry the key)
            meta[name.group(1)] = (get('vgpr_count'), get('vgpr_spill_count'), get('private_segment_fixed_size'), get('group_segment_fixed_size'))
    names = subprocess.run([cxxfilt], input='\n'.join(meta), capture_output=True, text=True, check=True).stdout.splitlines()
    by_name = {re.sub(r'\(.*$', '', re.sub(r'^void (pq::)?', '', d)): v for d, v in zip(names, meta.values())}

    def one(fragment):
        hits = [(k, v) for k, v in by_name.items() if fragment in k]
        assert len(hits) == 1, (fragment, [k for k, _ in hits])
        return hits[0][1]

    for kern in ('mfma_bgemm16_kernel<true>', 'mfma_bgemm16t_kernel<true>'):
        vgpr, spills, scratch, lds = one(kern)
        assert vgpr <= 128 and spills == 0 and scratch == 0, (kern, vgpr, spills, scratch)      # 512 / 128 = four waves per SIMD
        assert lds < 0 or 4 * lds <= 160 * 1024, (kern, lds)                                     # and four workgroups' tiles in a CU's LDS (36 KiB each)
    small, big = one('train_attn_dec_bf16_kernel<true, 2>'), one('train_attn_dec_bf16_kernel<true, 8>')
    assert small[0] <= 128 and small[1] == 0 and small[2] == 0, small
    assert big[1] == 0 and big[2] == 0 and big[0] <= 512, big


@pytest.mark.gpu
def test_plan_arena_through_the_callers_allocator(monkeypatch):
    """parseq_plan_create_ex (ABI 7): the plan's ONE device arena comes from the caller's allocator — here torch's caching allocator through
    parseq_amd._native.TorchPlanAllocator — is asked for exactly once per plan with exactly parseq_plan_workspace_bytes, the forward on it equals the
    forward on a hipMalloc'ed arena bit for bit, and every block goes back when the plans are destroyed.  alloc without release is refused."""
    import ctypes as C

    import torch
    from gpu_util import DEV, make_model
    from oracle.synth import CONFIGS, synth_images
    from parseq_amd import _native
    x = synth_images(16, CONFIGS['parseq'], seed=5).to(DEV)
    monkeypatch.setenv('PARSEQ_PLAN_ALLOCATOR', 'hip')        # the opt-out: a hipMalloc outside torch's allocator
    with torch.inference_mode():
        want = make_model('parseq', 'bf16x3')(x, 25).float().clone()
    monkeypatch.delenv('PARSEQ_PLAN_ALLOCATOR')               # the default (round 6): torch's caching allocator
    m = make_model('parseq', 'bf16x3')
    before = torch.cuda.memory_allocated()
    with torch.inference_mode():
        got = m(x, 25).float().clone()
        got2 = m(x, 25, slot=1).float().clone()      # a second plan (workspace slot) through the same allocator object
    torch.cuda.synchronize()
    st = m.model._native_state
    alloc = st.allocator
    lib = _native.lib()
    sizes = sorted(lib.parseq_plan_workspace_bytes(plan) for plan, _ in st.plans.values())
    assert alloc.calls == len(st.plans) == 2 and sorted(t.numel() for t in alloc.blocks.values()) == sizes
    assert alloc.bytes_out == sum(sizes) and torch.cuda.memory_allocated() - before >= sum(sizes)      # torch's allocator sees the workspace
    assert torch.equal(got, want) and torch.equal(got2, want)
    # the pair goes together
    handle = C.c_void_p(0)
    assert lib.parseq_plan_create_ex(st.model, 8, _native.PARSEQ_BF16, _native.stream_ptr(x), alloc.alloc_ptr, None, None, C.byref(handle)) != 0 and not handle.value
    st.release()
    assert alloc.bytes_out == 0 and not alloc.blocks


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['bf16x3', 'bf16', 'fp32'])
def test_plan_arena_recycled_from_a_poisoned_block(precision, monkeypatch):
    """The caller's allocator hands out RECYCLED memory: a caching allocator's block holds whatever its previous owner left (NaNs, old
    activations).  Nothing in a plan may rely on a zero-initialised arena — padding rows of ragged row tiles, partially written tables,
    counters.  The allocator handed to parseq_plan_create_ex here fills every arena with 0xFF bytes (NaN as f32 and as bf16, -1 as an
    integer) before the library sees it; every decode mode on a ragged batch must equal the run on a hipMalloc'ed arena bit for bit."""
    import torch
    from gpu_util import DEV, make_model
    from oracle.synth import CONFIGS, synth_images
    from parseq_amd import _native
    x = synth_images(13, CONFIGS['parseq'], seed=6).to(DEV)
    cases = [(True, 1, 25), (True, 0, None), (False, 2, None), (True, 2, 7)]

    def run(m, **kw):
        outs = []
        for ar, ri, ml in cases:
            m.model.decode_ar, m.model.refine_iters = ar, ri
            with torch.inference_mode():
                outs.append(m(x, ml, **kw).float().clone())
        torch.cuda.synchronize()
        return outs

    class PoisonedBlocks(_native.TorchPlanAllocator):
        def _do_alloc(self, nbytes, _user):
            ptr = super()._do_alloc(nbytes, _user)
            if ptr:
                self.blocks[ptr].fill_(0xFF)
                torch.cuda.synchronize()
            return ptr
    monkeypatch.setenv('PARSEQ_PLAN_ALLOCATOR', 'hip')
    ref = make_model('parseq', precision)
    want, want_slot = run(ref), run(ref, slot=1)
    monkeypatch.delenv('PARSEQ_PLAN_ALLOCATOR')
    m = make_model('parseq', precision)
    st = m.model._sync_native()
    st.allocator = PoisonedBlocks(DEV)
    got, got_slot = run(m), run(m, slot=1)
    assert st.allocator.calls == len(st.plans) == 2          # both arenas came through the poisoning allocator
    for a, b in zip(got + got_slot, want + want_slot):
        assert not torch.isnan(a).any() and torch.equal(a, b)
