"""Cycle stamps of one work item of mjtjinv_kernel / condense_kernel (PROF build: make -C robotoc_amd/csrc PROF=1
OUT=../librtoc_hip_prof.so BUILD=build_prof; RTOC_HIP_LIB=.../librtoc_hip_prof.so).  Usage: condense_profile.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_KKT, joint_limit_rows
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
capi.debug_profile(ctx)
ctx.condense(); ctx.sync()
ctx.upload(BUF_KKT, tile(kkt)); ctx.upload(BUF_CDD, tile(cdd))
ctx.condense(); ctx.sync()
p = capi.debug_profile(ctx).astype(np.int64).reshape(-1)[:32]
print("condense %.3f ms (best of 5: %.3f)" % (ctx.time_phase(2, 3), min(ctx.time_phase(2, 1) for _ in range(5))))
fused = os.environ.get("RTOC_CONDENSE_SPLIT", "1") == "0"
for name, slots in ((("fused condense_kernel (wave 0)", (0, 1, 2, 16, 17, 18, 19, 20, 3, 4, 23, 5, 6, 10, 11, 12, 13, 7, 8, 9)),) if fused else
                    (("mjtjinv_kernel", (14, 25, 26, 27, 28, 15, 1, 2, 3, 16, 17, 18, 19, 20, 24)), ("condense_kernel", (0, 21, 22, 23, 4, 5, 6, 10, 11, 12, 13, 7, 8, 9)))):
    vals = [(k, p[k]) for k in slots if p[k]]
    if vals:
        t0 = vals[0][1]
        print(name, " ".join("%d:%d" % (k, v - t0) for k, v in vals))
ctx.close()
