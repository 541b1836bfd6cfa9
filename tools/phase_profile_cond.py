"""Cycle stamps of one work item of condense_rv_kernel (PROF build: make -C robotoc_amd/csrc PROF=1 OUT=../librtoc_hip_prof.so
BUILD=build_prof; RTOC_HIP_LIB=.../librtoc_hip_prof.so).  Usage: phase_profile_cond.py [batch] [norows]"""
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
if "norows" not in sys.argv:
    ctx.set_constraint_rows(joint_limit_rows(dims))
    ctx.set_friction_cones(4, 3)
tile = lambda a: np.ascontiguousarray(np.tile(a, (batch // a.shape[0] + 1,) + (1,) * (a.ndim - 1))[:batch])
kkt, cdd = pr.make_precondense_batch_unique(L, grids, 32)
ctx.upload(BUF_KKT, tile(kkt)); ctx.upload(BUF_CDD, tile(cdd))
ctx.upload(BUF_CON, tile(pr.make_constraint_batch_unique(L, grids, 32)))
ctx.upload(BUF_CONE, tile(pr.make_cone_batch_unique(L, grids, 32, 4)))
ctx.set_condense_register(True)
capi.debug_profile(ctx)
ctx.condense(); ctx.sync()
ctx.upload(BUF_KKT, tile(kkt)); ctx.upload(BUF_CDD, tile(cdd))
ctx.condense(); ctx.sync()
p = capi.debug_profile(ctx).astype(np.int64).reshape(-1)
names = {0: "start", 3: "MJtJinv assembled + stored", 4: "operands in registers", 5: "LD", 6: "Xn, WL, staging synced", 7: "staging stores issued", 10: "V (Qxx, lx, hx)",
         11: "V2 (Qxu)", 12: "QU (Quu, lu)", 9: "end"}
t0 = p[64]
prev = t0
for k in (0, 3, 4, 5, 6, 7, 10, 11, 12, 9):
    v = p[64 + k]
    if v:
        print("%-32s %7d  (+%d)" % (names[k], v - t0, v - prev))
        prev = v
fn = {1: "inputs in LDS", 2: "LLT(M) + L^-1", 3: "M^-1 = Y^T Y", 16: "J M^-1", 17: "S = J M^-1 J^T + damping", 18: "LLT(S) + inverse factor", 19: "-S^-1",
      20: "topRight"}
prev = t0
for k in (1, 2, 3, 16, 17, 18, 19, 20):
    v = p[64 + 32 + k]
    if v:
        print("  fragment: %-28s %7d  (+%d)" % (fn[k], v - t0, v - prev))
        prev = v
print("condense %.3f ms" % min(ctx.time_phase(2, 1) for _ in range(5)))
ctx.close()
