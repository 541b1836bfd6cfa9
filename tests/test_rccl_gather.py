"""rtoc_gather_directions on real RCCL: a ONE-rank communicator on the single GPU of the test box
exercises the run-time binding (dlopen / dlsym of librccl, the ncclDataType constant, the stream hand-over) and the
collective itself; with one rank the all-gather must reproduce RTOC_BUF_DIR exactly.  (The 2/4/8-GPU path is the
same call with a larger communicator; its host-side sharding logic is covered by tests/test_sharding_gloo.py.)"""
import ctypes as C
import os

import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import BUF_DIR, BUF_DX0, BUF_KKT

pytestmark = pytest.mark.gpu


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]  # ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)


def _rccl():
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            return C.CDLL(name, mode=os.RTLD_GLOBAL)
        except OSError:
            continue
    raise RuntimeError("librccl not found")


def test_gather_directions_single_rank_rccl():
    import torch
    from robotoc_amd import capi
    lib = _rccl()
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    uid = _UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    dims, grids, _ = pr.config_anymal_trot()
    batch = 8
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.upload(BUF_KKT, pr.make_kkt_batch(L, grids, batch))
        ctx.upload(BUF_DX0, pr.make_dx0(L, batch))
        ctx.riccati_sweep()
        n = ctx.buffer_count(BUF_DIR)
        out = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        rc = capi.lib().rtoc_gather_directions(ctx._h, comm, C.cast(out.data_ptr(), C.POINTER(C.c_double)))
        assert rc == 0, capi.lib().rtoc_error_string(rc)
        ctx.sync()
        torch.cuda.synchronize()
        d = ctx.download_records(BUF_DIR, "dir")
        got = out.cpu().numpy().reshape(d.shape)
        assert np.isfinite(d).all() and np.abs(d).max() > 0
        assert np.array_equal(got, d)
        # bad arguments are API errors, not crashes
        assert capi.lib().rtoc_gather_directions(ctx._h, None, C.cast(out.data_ptr(), C.POINTER(C.c_double))) != 0
    finally:
        ctx.close()
        lib.ncclCommDestroy(comm)
