"""Cycle stamps of one work item of expand_kernel (PROF build; RTOC_HIP_LIB=.../librtoc_hip_prof.so).  Usage: expand_profile.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_KKT, BUF_DX0, joint_limit_rows
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims, grids, _ = pr.config_anymal_trot()
ctx = capi.Context(dims, len(grids), batch, 0)
L = ctx.L
ctx.set_grid(grids)
ctx.set_constraint_rows(joint_limit_rows(dims))
ctx.set_friction_cones(4, 3)
tile = lambda a: np.ascontiguousarray(np.tile(a, (batch // a.shape[0] + 1,) + (1,) * (a.ndim - 1))[:batch])
kkt, cdd = pr.make_precondense_batch_unique(L, grids, 32)
ctx.upload(BUF_KKT, tile(kkt)); ctx.upload(BUF_CDD, tile(cdd))
ctx.upload(BUF_CON, tile(pr.make_constraint_batch_unique(L, grids, 32)))
ctx.upload(BUF_CONE, tile(pr.make_cone_batch_unique(L, grids, 32, 4)))
ctx.upload(BUF_DX0, tile(pr.make_dx0_unique(L, 32)))
capi.debug_profile(ctx)
ctx.condense(); ctx.riccati_backward(); ctx.riccati_forward(); ctx.expand(0.995); ctx.sync()
ctx.expand(0.995); ctx.sync()
p = capi.debug_profile(ctx).astype(np.int64).reshape(-1)[:64]
print("expand %.3f ms (best of 5: %.3f)" % (ctx.time_phase(3, 3), min(ctx.time_phase(3, 1) for _ in range(5))))
vals = [(k, p[k]) for k in range(32, 40) if p[k]]
print("expand_kernel", " ".join("%d:%d" % (k, v - vals[0][1]) for k, v in vals))
ctx.close()
