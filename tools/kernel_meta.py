#!/usr/bin/env python3
"""Registers / spills / scratch / LDS of the kernels in libparseq_hip.so whose (mangled) name contains any of the given fragments.
Usage: python tools/kernel_meta.py enc_blocks_x3 ar24 [--lib path]"""
import re
import struct
import subprocess
import sys

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'


def code_objects(lib):
    blob = open(lib, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    at, out = blob.find(magic), []
    while at >= 0:
        n = struct.unpack_from('<Q', blob, at + len(magic))[0]
        off = at + len(magic) + 8
        for _ in range(n):
            o, size, tlen = struct.unpack_from('<QQQ', blob, off)
            triple = blob[off + 24:off + 24 + tlen].decode()
            off += 24 + tlen
            if 'gfx950' in triple and size:
                out.append(blob[at + o:at + o + size])
        at = blob.find(magic, at + len(magic))
    return out


def main():
    args = sys.argv[1:]
    lib = 'parseq_amd/lib/libparseq_hip.so'
    if '--lib' in args:
        lib = args[args.index('--lib') + 1]
        args = [a for a in args if a not in ('--lib', lib)]
    for i, co in enumerate(code_objects(lib)):
        path = f'/tmp/_km{i}.co'
        open(path, 'wb').write(co)
        notes = subprocess.run([READELF, '--notes', path], capture_output=True, text=True).stdout
        for blk in re.split(r'\n\s+- ', notes):
            nm = re.search(r'\.name:\s+(\S+)', blk)
            if nm and '.vgpr_count' in blk and any(f in nm.group(1) for f in args):
                g = lambda key: (re.search(r'\.%s:\s+(\d+)' % key, blk) or [0, '-'])[1]      # noqa: E731
                print(f"{nm.group(1)[:90]}: vgpr {g('vgpr_count')} agpr {g('agpr_count')} spill {g('vgpr_spill_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")


if __name__ == '__main__':
    main()
