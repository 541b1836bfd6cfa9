#!/usr/bin/env python3
"""Headline backward sweep (4096 distinct ANYmal trot instances) with the role-split and with the register-resident kernel
(RTOC_OPT_BACKWARD_REGISTER), timed with events on the context's stream; the two results compared on the device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_RIC, BUF_DX0, BUF_DIR

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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
res = {}
for name, on in (("role-split", False), ("register", True), ("role-split", False), ("register", True)):
    ctx.set_backward_register(on)
    ric.fill_(float("nan"))
    torch.cuda.synchronize()   # (torch's stream, not the context's)
    ctx.clear_status()
    for _ in range(5):
        ctx.riccati_backward()
    ctx.sync()
    t = [ctx.time_phase(0, 1) for _ in range(20)]
    bad = int((ctx.status() != 0).sum())
    res[name] = ric.clone()
    print("%-11s backward ms: min %.3f median %.3f  status!=0: %d" % (name, min(t), sorted(t)[len(t) // 2], bad), flush=True)
a, b = res["role-split"], res["register"]
P = slice(0, L.ric.off[2])
den = a[:, :, P].abs().amax(dim=2).clamp_min(1e-300)
err = ((a[:, :, P] - b[:, :, P]).abs().amax(dim=2) / den)
print("P, s of the two kernels: worst relative difference %.3e at (instance, stage) %s" % (float(err.max()), tuple(int(v) for v in (err == err.max()).nonzero()[0])))
ctx.riccati_forward(); ctx.sync()
print("forward ms", min(ctx.time_phase(1, 1) for _ in range(5)))
ctx.close()
