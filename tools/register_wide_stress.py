"""Repeatability stress of the register-wide backward kernels (riccati_backward_rw.hpp, _rw2.hpp: the hand-over between the two waves
of an instance through LDS): 300 recursions over 1024 distinct iCub instances each, every Riccati record compared with the first run's
bit for bit on the device.  Usage: python tools/register_wide_stress.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC
for nv in (35, 32):
    dims, grids, _ = pr.config_icub_jump(nv=nv)
    batch, n = 1024, len(grids)
    ctx = capi.Context(dims, n, batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    ctx.set_fxx_structure(2)
    z = lambda w: torch.zeros((batch, n, getattr(L, w).stride), dtype=torch.float64, device="cuda:0")
    kkt = pr.make_kkt_batch_unique(L, grids, batch, seed=5, backend="torch", device="cuda:0", out=z("kkt"))
    dx0 = pr.make_dx0_unique(L, batch, seed=5, backend="torch", device="cuda:0").contiguous()
    ric, d = z("ric"), z("dir")
    for b_, t_ in ((BUF_KKT, kkt), (BUF_DX0, dx0), (BUF_RIC, ric), (BUF_DIR, d)):
        ctx.bind(b_, t_.data_ptr())
    torch.cuda.synchronize()
    first, bad = None, 0
    reps = 300
    for rep in range(reps):
        ric.fill_(float("nan"))
        torch.cuda.synchronize()   # (the fill runs on torch's stream, the recursion on the context's)
        ctx.riccati_backward()
        ctx.sync()
        assert int((ctx.status() != 0).sum()) == 0
        if first is None:
            first = ric.clone()
        else:
            ne = first.view(torch.int64) != ric.view(torch.int64)
            ne &= ~(torch.isnan(first) & torch.isnan(ric))
            if bool(ne.any()):
                bad += 1
                idx = ne.nonzero()
                print("nv", nv, "rep", rep, "words differ", int(ne.sum()), "instances", sorted(set(idx[:, 0].tolist()))[:6], "stages", int(idx[:, 1].min()), int(idx[:, 1].max()))
    print("nv=%d: %d repetitions of 1024 instances, %d differ from the first" % (nv, reps, bad))
    ctx.close()
