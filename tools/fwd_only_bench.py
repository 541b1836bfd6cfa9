#!/usr/bin/env python3
"""Forward recursion of the headline batch alone (after one backward sweep), min / median of 30 timings: compares builds of
riccati_forward.hpp with other FWD_HEAD / FWD_REST_PARTS (RTOC_HIP_LIB=...)."""
import os, sys
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
ctx.riccati_backward(); ctx.sync()
for _ in range(5):
    ctx.riccati_forward()
ctx.sync()
t = sorted(ctx.time_phase(1, 1) for _ in range(30))
print("forward ms: min %.4f median %.4f  (status != 0: %d)" % (t[0], t[15], int((ctx.status() != 0).sum())))
ctx.close()
