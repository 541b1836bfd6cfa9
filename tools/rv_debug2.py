#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_RIC, BUF_DX0, BUF_DIR
batch = 4096
dims, grids, _ = pr.config_anymal_trot()
n = len(grids)
ctx = capi.Context(dims, n, batch, 0)
L = ctx.L
ctx.set_grid(grids)
z = lambda w: torch.zeros((batch, n, getattr(L, w).stride), dtype=torch.float64, device="cuda:0")
kkt = pr.make_kkt_batch_unique(L, grids, batch, seed=0, backend="torch", device="cuda:0", out=z("kkt"))
dx0 = pr.make_dx0_unique(L, batch, seed=0, backend="torch", device="cuda:0").contiguous()
ric, d = z("ric"), z("dir")
for b_, t_ in ((BUF_KKT, kkt), (BUF_DX0, dx0), (BUF_RIC, ric), (BUF_DIR, d)):
    ctx.bind(b_, t_.data_ptr())
torch.cuda.synchronize()
def bits(label):
    st = torch.from_numpy(ctx.status().astype("int64"))
    print("%-40s" % label, {int(v): int((st == v).sum()) for v in st.unique()}, flush=True)
    ctx.clear_status()
ctx.set_backward_register(False)
ctx.riccati_backward(); ctx.sync(); ref = ric.clone(); bits("role-split x1")
ctx.set_backward_register(True)
ric.fill_(float("nan")); torch.cuda.synchronize()
ctx.riccati_backward(); ctx.sync(); bits("register x1 on NaN records")
ctx.riccati_backward(); ctx.sync(); bits("register x1 on its own records")
ric.copy_(ref); torch.cuda.synchronize()
ctx.riccati_backward(); ctx.sync(); bits("register x1 on role-split records")
for k in range(3):
    ctx.riccati_backward()
ctx.sync(); bits("register x3 back to back")
ric.zero_(); torch.cuda.synchronize()
ctx.riccati_backward(); ctx.sync(); bits("register x1 on zero records")
o = L.ric.off
P = slice(o[0], o[0] + 36 * 36)
err = ((ric[:, :, P] - ref[:, :, P]).abs().amax(dim=2) / ref[:, :, P].abs().amax(dim=2).clamp_min(1e-300))
err = torch.nan_to_num(err, nan=9.0).amax(dim=0)
print("P vs role-split per stage:", ["%d:%.1e" % (s, float(err[s])) for s in range(n - 1, -1, -1) if err[s] > 1e-9][:12])
ctx.close()
