#!/usr/bin/env python3
"""Debug aid for the register-resident backward kernel: status bits and per-stage differences against the role-split kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_RIC, BUF_DX0, BUF_DIR

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dims, grids, _ = pr.config_anymal_trot()
n = len(grids)
ctx = capi.Context(dims, n, batch, 0)
L = ctx.L
ctx.set_grid(grids)
z = lambda w: torch.zeros((batch, n, getattr(L, w).stride), dtype=torch.float64, device="cuda:0")
kkt = pr.make_kkt_batch_unique(L, grids, batch, seed=seed, backend="torch", device="cuda:0", out=z("kkt"))
dx0 = pr.make_dx0_unique(L, batch, seed=seed, backend="torch", device="cuda:0").contiguous()
ric, d = z("ric"), z("dir")
for b_, t_ in ((BUF_KKT, kkt), (BUF_DX0, dx0), (BUF_RIC, ric), (BUF_DIR, d)):
    ctx.bind(b_, t_.data_ptr())
torch.cuda.synchronize()
o = L.ric.off
def run(on, label):
    ctx.set_backward_register(on)
    ctx.clear_status()
    ric.fill_(float("nan"))
    torch.cuda.synchronize()
    ctx.riccati_backward(); ctx.sync()
    st = torch.from_numpy(ctx.status().astype("int64"))
    print("%-22s status bits:" % label, {int(v): int((st == v).sum()) for v in st.unique()}, flush=True)
    return ric.clone()
r1 = run(True, "register, run 1")
r2 = run(True, "register, run 2")
r0 = run(False, "role-split")
r3 = run(True, "register after rs")
for name, a in (("run1", r1), ("run2", r2), ("after-rs", r3)):
    P = slice(o[0], o[0] + 36 * 36)
    S = slice(o[1], o[1] + 36)
    nanP = torch.isnan(a[:, :, P]).any(dim=2)   # [batch, stage]
    print(name, "stages with NaN in P (count of instances):", {int(s): int(nanP[:, s].sum()) for s in range(n) if nanP[:, s].any()})
    den = r0[:, :, P].abs().amax(dim=2).clamp_min(1e-300)
    err = ((a[:, :, P] - r0[:, :, P]).abs().amax(dim=2) / den)
    err = torch.nan_to_num(err, nan=9.0)
    worst = err.amax(dim=0)
    print(name, "worst rel diff of P per stage vs role-split:", ["%d:%.1e" % (s, float(worst[s])) for s in range(n - 1, -1, -1) if worst[s] > 1e-9][:12])
    es = ((a[:, :, S] - r0[:, :, S]).abs().amax(dim=2) / r0[:, :, S].abs().amax(dim=2).clamp_min(1e-300))
    es = torch.nan_to_num(es, nan=9.0).amax(dim=0)
    print(name, "worst rel diff of s per stage:", ["%d:%.1e" % (s, float(es[s])) for s in range(n - 1, -1, -1) if es[s] > 1e-9][:12])
ctx.close()
