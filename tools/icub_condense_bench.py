import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_CDD
for nv in (32, 35):
    dims, grids, _ = pr.config_icub_jump(nv=nv)
    batch = 512
    ctx = capi.Context(dims, len(grids), batch, 0); L = ctx.L; ctx.set_grid(grids)
    k4, c4 = pr.make_precondense_batch(L, grids, 2)
    kkt = np.ascontiguousarray(np.tile(k4, (batch // 2, 1, 1))); cdd = np.ascontiguousarray(np.tile(c4, (batch // 2, 1, 1)))
    t = []
    for r in range(3):
        ctx.upload(BUF_KKT, kkt); ctx.upload(BUF_CDD, cdd)
        t.append(ctx.time_phase(2, 1))
    print("icub%d condense %.3f ms / %d" % (nv, min(t), batch), "status", int((ctx.status() != 0).sum()))
    ctx.close()
