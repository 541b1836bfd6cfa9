"""Backward-kernel timing of the iCub configurations (1024 instances): the default dispatch (the register-wide kernels: nv = 32
riccati_backward_rw.hpp, nv = 35 riccati_backward_rw2.hpp; owned buffers: no per-recursion check of the Fxx structure) and the tile-split kernel at every wave count compiled in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_DX0, BUF_KKT

def tile(a, batch):
    reps = (batch + a.shape[0] - 1) // a.shape[0]
    return np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:batch])

for nv, waves in ((32, (4,)), (35, (5, 4))):
    dims, grids, _ = pr.config_icub_jump(nv=nv)
    batch = 1024
    ctx = capi.Context(dims, len(grids), batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    ctx.upload(BUF_KKT, tile(pr.make_kkt_batch_unique(L, grids, 16, seed=7), batch))
    ctx.upload(BUF_DX0, tile(pr.make_dx0_unique(L, 16, seed=7), batch))
    ctx.time_phase(0, 3)
    print("iCub nv=%d, default dispatch: backward %.3f ms / %d instances (min of 5 x 3 launches), status nonzero %d" % (
        nv, min(ctx.time_phase(0, 3) for _ in range(5)), batch, int((ctx.status() != 0).sum())))
    ctx.set_backward_register(0)   # the tile-split kernel at every wave count compiled in
    for w in waves:
        ctx.set_backward_waves(w)
        ctx.time_phase(0, 1)
        print("iCub nv=%d, %d waves: backward %.3f ms / %d instances, status nonzero %d" % (
            nv, w, ctx.time_phase(0, 3), batch, int((ctx.status() != 0).sum())))
    ctx.close()
