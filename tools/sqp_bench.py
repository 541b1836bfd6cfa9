"""Quick SQP-phase timing (condense / backward / forward / expand) for tuning."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_CDD, BUF_DX0
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
acc = {}
for rep in range(3):
    kw[:kkt0.numel()].copy_(kkt0.view(-1)); cw[:cdd0.numel()].copy_(cdd0.view(-1)); torch.cuda.synchronize()
    for name, ph in (("condense", 2), ("backward", 0), ("forward", 1), ("expand", 3)):
        ms = ctx.time_phase(ph, 1)
        if rep: acc[name] = acc.get(name, 0) + ms / 2
print({k: round(v, 3) for k, v in acc.items()}, "total %.3f ms" % sum(acc.values()), "status!=0:", int((ctx.status() != 0).sum()))
if len(sys.argv) > 2:
    capi.debug_profile(ctx)
    kw[:kkt0.numel()].copy_(kkt0.view(-1)); cw[:cdd0.numel()].copy_(cdd0.view(-1)); torch.cuda.synchronize()
    ctx.condense(); ctx.sync()
    pp = capi.debug_profile(ctx).reshape(-1)
    print("KKT phase detail: Qxx %d | Qxup,Quuptr %d | Qxu,Quu %d | grad x %d | grad u %d" % (pp[10] - pp[6], pp[11] - pp[10], pp[12] - pp[11], pp[13] - pp[12], pp[7] - pp[13]))
    print("Lam phase detail: JMinv %d | S %d | LLT(S) %d | Sinv %d | TR %d | TL %d" % (pp[16] - pp[3], pp[17] - pp[16], pp[18] - pp[17], pp[19] - pp[18], pp[20] - pp[19], pp[4] - pp[20]))
    print("load phase detail: issue %d | zero fill %d | wait+store %d | barrier,cleanup %d" % (pp[21] - pp[0], pp[22] - pp[21], pp[23] - pp[22], pp[1] - pp[23]))
    p = pp[:10]
    names = ["load", "LLT(M)", "Minv", "J..Lam", "LD,Lr", "Qafqv/Qafu", "KKT updates", "dyn+SC+tail", "s2g"]
    print("condense item 0 ticks:", {n: int(x) for n, x in zip(names, np.diff(p))}, "total", int(p[9] - p[0]))
