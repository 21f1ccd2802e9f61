"""Builds parseq_amd/lib/libparseq_hip.so from parseq_amd/csrc with hipcc for gfx950 (cross-compiles without a GPU).

The shared library is the product: there is no pure-Python or CPU execution path.  It is built in-tree so that it
travels with the repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).

The library is eight translation units (UNITS) compiled in parallel and linked once; a unit is recompiled when the
SHA-256 of its source, of every header under csrc/, of include/parseq_hip.h and of the compiler flags differs from the one
recorded beside its object file (content, not mtime: a fresh checkout with an up-to-date library does not rebuild, an
edited header always does).  Wall time of a full build on 8 cores: under a minute (the longest unit ≈ 45 s); the single
translation unit it replaces took 2 min 10 s.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
OBJ_DIR = os.path.join(LIB_DIR, 'obj')          # git-ignored AND gpurun-ignored (as are the obj_<suffix> of A/B builds): only .so files travel
LIB_PATH = os.path.join(LIB_DIR, 'libparseq_hip.so')
STAMP_PATH = LIB_PATH + '.sha256'
UNITS = ['lib_model', 'lib_encode', 'lib_decode', 'lib_train', 'lib_ops', 'kern_enc_blocks', 'kern_enc_blocks_x3', 'kern_enc_blocks_x3w']
SOURCES = [u + '.hip' for u in UNITS]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'parseq_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed']
# per-unit flags on top of FLAGS: the two-waves-per-SIMD bf16x3 encoder keeps its accumulators in VGPRs (encoder_blocks_x3w.h)
UNIT_FLAGS = {'kern_enc_blocks_x3w': ['-mllvm', '-amdgpu-mfma-vgpr-form']}


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)')


def _extra_flags() -> list[str]:
    """PARSEQ_BUILD_FLAGS: extra compiler flags (e.g. -DX3_... experiment switches); part of every unit's hash."""
    return os.environ.get('PARSEQ_BUILD_FLAGS', '').split()


def _headers_digest() -> bytes:
    h = hashlib.sha256()
    for f in HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read() + b'\0')
    h.update(' '.join(FLAGS + _extra_flags() + [u + '=' + ' '.join(f) for u, f in sorted(UNIT_FLAGS.items())]).encode())
    return h.digest()


def _unit_hash(unit: str, headers: bytes) -> str:
    h = hashlib.sha256(headers)
    with open(os.path.join(CSRC, unit + '.hip'), 'rb') as fh:
        h.update(fh.read())
    return h.hexdigest()


def source_hash() -> str:
    """Hash of everything the library is built from."""
    headers = _headers_digest()
    return hashlib.sha256(''.join(_unit_hash(u, headers) for u in UNITS).encode()).hexdigest()


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as fh:
        return fh.read().strip() != source_hash()


def _compile(unit: str, want: str, verbose: bool) -> None:
    obj = os.path.join(OBJ_DIR, unit + '.o')
    stamp = obj + '.sha256'
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return
    tmp = f'{obj}.tmp{os.getpid()}'
    cmd = [_hipcc()] + FLAGS + UNIT_FLAGS.get(unit, []) + _extra_flags() + ['-c', os.path.join(CSRC, unit + '.hip'), '-o', tmp]
    if verbose:
        print('[parseq_amd.build]', ' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, obj)
    _replace_text(stamp, want)


def build(force: bool = False, verbose: bool = True) -> str:
    """Builds the library if it is stale.  Safe under several processes that import at once (torchrun ranks): the whole build runs under an exclusive flock on
    lib/.build.lock — the first rank builds, the others find the library fresh when they get the lock — and objects, stamps and the library are written to temporary
    names and os.replace()d into place, so that nobody ever dlopens or links a half-written file."""
    if not force and not is_stale():
        return LIB_PATH
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():      # another process built it while this one waited for the lock
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _replace_text(path: str, text: str) -> None:
    tmp = f'{path}.tmp{os.getpid()}'
    with open(tmp, 'w') as fh:
        fh.write(text)
    os.replace(tmp, path)


def _build_locked(force: bool, verbose: bool) -> str:
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    headers = _headers_digest()
    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 1)) as pool:
        for fut in [pool.submit(_compile, u, _unit_hash(u, headers), verbose) for u in UNITS]:
            fut.result()
    tmp = f'{LIB_PATH}.tmp{os.getpid()}'
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + [os.path.join(OBJ_DIR, u + '.o') for u in UNITS]
    if verbose:
        print('[parseq_amd.build]', ' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB_PATH)
    _replace_text(STAMP_PATH, source_hash())
    return LIB_PATH


def build_variant(suffix: str, flags: list[str], units: list[str] | None = None) -> str:
    """A/B builds: parseq_amd/lib/libparseq_hip_<suffix>.so with extra -D flags; only `units` (default: all) are recompiled with
    the flags, the other objects are the product build's.  Load it with PARSEQ_HIP_LIB=<path>."""
    build(verbose=False)
    units = units or UNITS
    vdir = os.path.join(LIB_DIR, 'obj_' + suffix)
    os.makedirs(vdir, exist_ok=True)

    def one(u: str) -> None:
        subprocess.run([_hipcc()] + FLAGS + UNIT_FLAGS.get(u, []) + flags + ['-c', os.path.join(CSRC, u + '.hip'), '-o', os.path.join(vdir, u + '.o')], check=True)
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
        for fut in [pool.submit(one, u) for u in units]:
            fut.result()
    out = os.path.join(LIB_DIR, f'libparseq_hip_{suffix}.so')
    objs = [os.path.join(vdir if u in units else OBJ_DIR, u + '.o') for u in UNITS]
    subprocess.run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, check=True)
    return out


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--variant':      # python -m parseq_amd.build --variant <suffix> [--units a,b] -DX=1 ...
        rest = sys.argv[3:]
        units = None
        if rest and rest[0] == '--units':
            units, rest = rest[1].split(','), rest[2:]
        print(build_variant(sys.argv[2], rest, units))
    else:
        print(build(force='--force' in sys.argv))
