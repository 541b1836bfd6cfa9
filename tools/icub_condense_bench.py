"""Condensation of the iCub configurations (1024 instances, joint-limit rows + 2 x 17 wrench-cone rows) on both pipelines:
RTOC_OPT_CONDENSE_SPLIT = 1 (mjtjinv_kernel + condense_kernel) and 0 (the one-kernel form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_CDD, BUF_CON, BUF_CONE, icub_dims, joint_limit_rows
for nv in (32, 35):
    d0, grids, _ = pr.config_icub_jump(nv=nv)
    dims = icub_dims(nv, nc_max=(6 * d0.nu + 34 + 7) & ~7)
    batch = 1024
    ctx = capi.Context(dims, len(grids), batch, 0); L = ctx.L; ctx.set_grid(grids)
    ctx.set_constraint_rows(joint_limit_rows(dims)); ctx.set_wrench_cones(2)
    tile = lambda a: np.ascontiguousarray(np.tile(a, (batch // a.shape[0],) + (1,) * (a.ndim - 1)))
    k4, c4 = pr.make_precondense_batch(L, grids, 4)
    kkt, cdd, con = tile(k4), tile(c4), tile(pr.make_constraint_batch(L, grids, 4))
    cones = [capi.wrench_cone_matrix(0.1, 0.05, 0.6), capi.wrench_cone_matrix(0.09, 0.055, 0.7)]
    ctx.upload(BUF_CONE, pr.make_wrench_cone_batch(L, grids, batch, 2, cones))
    for split in (1, 0, 1, 0):
        ctx.set_condense_split(split)
        t = []
        for r in range(3):
            ctx.upload(BUF_KKT, kkt); ctx.upload(BUF_CDD, cdd); ctx.upload(BUF_CON, con)
            t.append(ctx.time_phase(2, 1))
        print("icub%d split=%d condense %.3f ms / %d" % (nv, split, min(t), batch), "status", int((ctx.status() != 0).sum()))
    ctx.close()
