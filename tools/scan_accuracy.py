import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import oracle
from robotoc_amd import capi, grid as G, problems as pr
from robotoc_amd.types import BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, Records, anymal_dims
from helpers import rel_err
dims = anymal_dims()
for N in (20, 40, 63, 100):
    grids = G.uniform_grid(N, 0.02, dimf=12)
    for mode in ("dynamics", "factory"):
        out = {}
        for scan in (False, True):
            ctx = capi.Context(dims, len(grids) + 1, 1, 0)
            L = ctx.L
            ctx.set_grid(grids); ctx.set_backward_scan(scan)
            kkt = pr.make_kkt_batch(L, grids, 1, mode=mode); dx0 = pr.make_dx0(L, 1)
            ctx.upload(BUF_KKT, kkt); ctx.upload(BUF_DX0, dx0)
            ctx.riccati_backward(); ctx.riccati_forward()
            out[scan] = (ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir"))
            ctx.close()
        ric_ref = Records(L, "ric").zeros(1, len(grids)); d_ref = Records(L, "dir").zeros(1, len(grids))
        oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        R, D = Records(L, "ric"), Records(L, "dir")
        res = []
        for scan in (False, True):
            eP = max(rel_err(R.f(out[scan][0][0, i], "P"), R.f(ric_ref[0, i], "P")) for i in range(len(grids)))
            eK = max(rel_err(R.f(out[scan][0][0, i], "K"), R.f(ric_ref[0, i], "K")) for i in range(len(grids) - 1))
            ex = max(rel_err(D.f(out[scan][1][0, i], "dx"), D.f(d_ref[0, i], "dx")) for i in range(len(grids)))
            el = max(rel_err(D.f(out[scan][1][0, i], "dlmdgmm"), D.f(d_ref[0, i], "dlmdgmm")) for i in range(len(grids)))
            res.append("%s: P %.1e K %.1e dx %.1e dlmd %.1e" % ("scan  " if scan else "serial", eP, eK, ex, el))
        print("N=%d %s | %s | %s" % (N, mode, res[0], res[1]))
