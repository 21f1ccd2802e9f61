"""INTEGRATION.md's recipe for a caller WITHOUT torch.distributed, executed: one process per GPU, the caller's own RCCL communicator and
stream, `parseq_shard_bounds` for the split, `parseq_forward` on this rank's crops, ONE `ncclAllGather` of the logits on the same stream —
no host synchronisation between the forward and the collective.  RCCL is driven directly through ctypes (the librccl.so that ships with
torch is only the FILE used; no process group exists).  One rank is what a 1-GPU box can run: the split, the communicator, the call
sequence, the stream ordering and the buffer arithmetic are the recipe's; the wire is not exercised."""
import ctypes as C
import glob
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


def _rccl():
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so*')) + glob.glob('/opt/rocm/lib/librccl.so*')
    if not cands:
        pytest.skip('no librccl.so on this box')
    lib = C.CDLL(cands[0])
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllGather, lib.ncclCommDestroy):
        f.restype = C.c_int
    return lib


def test_shard_forward_allgather_without_torch_distributed(golden):
    from gpu_util import DEV, make_model
    from parseq_amd import _native
    assert not (torch.distributed.is_available() and torch.distributed.is_initialized())
    rccl = _rccl()
    g, _ = golden('parseq')
    m = make_model('parseq', 'bf16x3')
    images = g['images'].to(DEV).repeat(9, 1, 1, 1).contiguous()            # 72 crops
    n_crops, world, rank, L, Cc = images.shape[0], 1, 0, 26, 95
    with torch.inference_mode():
        want = m(images, 25).float().clone()                                   # the same call through the Python surface
    torch.cuda.synchronize()
    lib = _native.lib()
    uid, comm = _UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    try:
        b0, b1 = C.c_int64(), C.c_int64()
        _native.check(lib.parseq_shard_bounds(n_crops, world, rank, C.byref(b0), C.byref(b1)))
        assert (b0.value, b1.value) == (0, n_crops)
        rows = b1.value - b0.value
        stream = torch.cuda.Stream()
        local = torch.empty(rows, L, Cc, dtype=torch.float32, device=DEV)
        gathered = torch.zeros(world * rows, L, Cc, dtype=torch.float32, device=DEV)
        out_len = C.c_int(0)
        with torch.cuda.stream(stream):
            plan = m.model._plan(rows, 0)
            sp = C.c_void_p(stream.cuda_stream)
            _native.check(lib.parseq_forward(plan, C.c_void_p(images.data_ptr() + b0.value * 3 * 32 * 128 * 4), _native.dtype_code(images.dtype), rows,
                                             _native.FLAG_DECODE_AR, 1, L, _native.ptr(local), C.byref(out_len), sp))
            assert out_len.value == L                                          # refine_iters >= 1: every rank returns all 26 positions
            assert rccl.ncclAllGather(_native.ptr(local), _native.ptr(gathered), rows * L * Cc, 7, comm, sp) == 0      # 7 = ncclFloat32
        stream.synchronize()
        assert torch.equal(gathered, want)
    finally:
        rccl.ncclCommDestroy(comm)
