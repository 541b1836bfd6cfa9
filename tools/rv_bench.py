#!/usr/bin/env python3
"""Backward sweep of a batch of distinct instances with RTOC_OPT_BACKWARD_REGISTER off and on (the register-resident kernels where
they apply), interleaved A B A B, timed with events on the context's stream; the two results compared on the device.
usage: rv_bench.py [batch] [trot | jump_sto | icub32 | icub35]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_RIC, BUF_DX0, BUF_DIR

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = sys.argv[2] if len(sys.argv) > 2 else "trot"
dims, grids, _ = {"trot": pr.config_anymal_trot, "jump_sto": pr.config_anymal_jump_sto, "icub32": lambda: pr.config_icub_jump(nv=32),
                  "icub35": lambda: pr.config_icub_jump(nv=35)}[cfg]()
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
if os.environ.get("RTOC_FXX"):   # 2: the caller asserts the structure of Fxx (no in-kernel verification of a bound buffer)
    ctx.set_fxx_structure(int(os.environ["RTOC_FXX"]))
res = {}
for name, on in (("role-split", 0), ("register", 2), ("role-split", 0), ("register", 2)):
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
den = torch.nan_to_num(a[:, :, P]).abs().amax(dim=2).clamp_min(1e-300)
err = (torch.nan_to_num(a[:, :, P] - b[:, :, P]).abs().amax(dim=2) / torch.nan_to_num(den, nan=1.0))   # (padding between fields stays NaN in both)
print("P, s of the two kernels: worst relative difference %.3e at (instance, stage) %s" % (float(err.max()), tuple(int(v) for v in (err == err.max()).nonzero()[0])))
ctx.riccati_forward(); ctx.sync()
print("forward ms", min(ctx.time_phase(1, 1) for _ in range(5)))
ctx.close()
