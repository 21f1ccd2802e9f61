#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in libparseq_hip.so, read from the gfx950 code object embedded in the
library (no GPU needed): unbundles the __CLANG_OFFLOAD_BUNDLE__ in .hip_fatbin and prints the AMDGPU metadata note.

    python tools/kernel_resources.py [pattern ...]      # only kernels whose name contains one of the patterns
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get('PARSEQ_HIP_LIB', os.path.join(ROOT, 'parseq_amd', 'lib', 'libparseq_hip.so'))
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
FILT = '/usr/bin/c++filt'


def code_object(path):
    blob = open(path, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    at = blob.find(magic)
    if at < 0:
        raise SystemExit('no offload bundle in ' + path)
    n = struct.unpack_from('<Q', blob, at + len(magic))[0]
    off = at + len(magic) + 8
    for _ in range(n):
        o, size, tlen = struct.unpack_from('<QQQ', blob, off)
        triple = blob[off + 24:off + 24 + tlen].decode()
        off += 24 + tlen
        if 'gfx950' in triple:
            return blob[at + o:at + o + size]
    raise SystemExit('no gfx950 entry in the bundle')


def main():
    pats = sys.argv[1:]
    with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
        f.write(code_object(LIB))
        co = f.name
    try:
        notes = subprocess.run([READELF, '--notes', co], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(co)
    rows = []
    for blk in re.split(r'\n\s+- ', notes):
        name = re.search(r'\.name:\s+(\S+)', blk)
        if not name or '.vgpr_count' not in blk:
            continue
        get = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]      # noqa: E731
        rows.append((name.group(1), get('vgpr_count'), get('agpr_count'), get('sgpr_count'), get('vgpr_spill_count'),
                     get('private_segment_fixed_size'), get('group_segment_fixed_size'), get('max_flat_workgroup_size')))
    names = subprocess.run([FILT], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print('| kernel | vgpr | agpr | sgpr | vgpr spills | scratch B | static LDS B | max wg |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    for r, nm in sorted(zip(rows, names), key=lambda t: t[1]):
        nm = re.sub(r'^void (pq::)?', '', nm)
        nm = re.sub(r'\(.*$', '', nm)
        if pats and not any(p in nm for p in pats):
            continue
        print(f'| `{nm}` | ' + ' | '.join(r[1:]) + ' |')


if __name__ == '__main__':
    main()
