"""Backward / forward recursion of the batch configurations (HIP events on the launch stream), for tuning runs:
ANYmal trot, ANYmal jump with STO (4096 instances), iCub nv = 32 / 35 (1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0
for name, fn, batch in (("anymal_trot", pr.config_anymal_trot, 4096), ("anymal_jump_sto", pr.config_anymal_jump_sto, 4096),
                        ("icub32", lambda: pr.config_icub_jump(nv=32), 1024), ("icub35", lambda: pr.config_icub_jump(nv=35), 1024)):
    dims, grids, _ = fn()
    ctx = capi.Context(dims, len(grids), batch, 0)
    ctx.set_grid(grids)
    ctx.upload(BUF_KKT, pr.make_kkt_batch_tiled(ctx.L, grids, batch, unique=8))
    ctx.upload(BUF_DX0, np.tile(pr.make_dx0(ctx.L, 8), (batch // 8 + 1, 1))[:batch])
    ctx.time_phase(0, 3)
    b = min(ctx.time_phase(0, 10) for _ in range(3))
    f = min(ctx.time_phase(1, 10) for _ in range(3))
    print("%-16s batch %d: backward %.3f ms, forward %.3f ms, status nonzero %d" % (name, batch, b, f, int((ctx.status() != 0).sum())))
    ctx.close()
