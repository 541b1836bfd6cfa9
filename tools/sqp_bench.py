"""SQP-phase timing with joint-limit AND friction-cone rows (bench.py's sqp_iteration leg)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_CDD, BUF_CON, BUF_CONE, joint_limit_rows
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims, grids, _ = pr.config_anymal_trot()
ctx = capi.Context(dims, len(grids), batch, 0); L = ctx.L; ctx.set_grid(grids)
k4, c4 = pr.make_precondense_batch(L, grids, 4)
rp = (batch + 3) // 4
kkt0 = torch.from_numpy(np.ascontiguousarray(np.tile(k4, (rp, 1, 1))[:batch])).cuda()
cdd0 = torch.from_numpy(np.ascontiguousarray(np.tile(c4, (rp, 1, 1))[:batch])).cuda()
kw = torch.empty(ctx.buffer_count(BUF_KKT), dtype=torch.float64, device="cuda")
cw = torch.empty(ctx.buffer_count(BUF_CDD), dtype=torch.float64, device="cuda")
ctx.bind(BUF_KKT, kw.data_ptr()); ctx.bind(BUF_CDD, cw.data_ptr())
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_constraint_rows(joint_limit_rows(dims))
ctx.set_friction_cones(4, 3)
ctx.upload(BUF_CONE, np.ascontiguousarray(np.tile(pr.make_cone_batch(L, grids, 4, 4), (rp, 1, 1))[:batch]))
con0 = torch.from_numpy(np.ascontiguousarray(np.tile(pr.make_constraint_batch(L, grids, 4), (rp, 1, 1))[:batch])).cuda()
nw = torch.empty(ctx.buffer_count(BUF_CON), dtype=torch.float64, device="cuda")
ctx.bind(BUF_CON, nw.data_ptr())
acc = {}
for rep in range(3):
    kw[:kkt0.numel()].copy_(kkt0.view(-1)); cw[:cdd0.numel()].copy_(cdd0.view(-1)); nw[:con0.numel()].copy_(con0.view(-1)); torch.cuda.synchronize()
    for name, ph in (("condense", 2), ("backward", 0), ("forward", 1), ("expand", 3), ("update", 5)):
        ms = ctx.time_phase(ph, 1)
        if rep: acc[name] = acc.get(name, 0) + ms / 2
print({k: round(v, 3) for k, v in acc.items()}, "total %.3f ms" % sum(acc.values()), "status!=0:", int((ctx.status() != 0).sum()))
