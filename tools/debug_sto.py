import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle as orc
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import *
from helpers import rel_err
dims, grids, _ = pr.config_anymal_jump_sto()
import itertools
for batch, extra, setd in [(1,0,False),(4,0,False),(1,2,False),(1,0,True)]:
    nw=1
    print('CASE batch',batch,'extra',extra,'setd',setd)
    ctx = capi.Context(dims, len(grids)+extra, batch, 0)
    L = ctx.L; ctx.set_grid(grids); ctx.set_backward_waves(nw)
    if setd: ctx.set_max_dts0(0.1)
    kkt = pr.make_kkt_batch(L, grids, batch); dx0 = pr.make_dx0(L, batch)
    ctx.upload(BUF_KKT, kkt); ctx.upload(BUF_DX0, dx0)
    ctx.riccati_backward(); ctx.riccati_forward()
    ric = ctx.download_records(BUF_RIC, "ric"); d = ctx.download_records(BUF_DIR, "dir")
    R = Records(L, "ric"); D = Records(L, "dir")
    ric_ref = R.zeros(batch, len(grids)); d_ref = D.zeros(batch, len(grids))
    orc.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
    for i in (25,):
        row = ["st %2d" % i]
        for f in ["P", "s", "K", "k", "M", "m", "mt", "mt_next", "Psi", "Phi", "T", "W", "scal", "dtsdx"]:
            row.append("%s %.1e" % (f, rel_err(R.f(ric[0, i], f), R.f(ric_ref[0, i], f))))
        for f in ["dx", "du", "dlmdgmm", "dxi", "dts"]:
            row.append("%s %.1e" % (f, rel_err(D.f(d[0, i], f), D.f(d_ref[0, i], f))))
        print(" ".join(row))
    i = 25
    print("dts gpu", D.f(d[0, i], "dts")[:2], "ref", D.f(d_ref[0, i], "dts")[:2])
    print("dxi gpu", D.f(d[0, i], "dxi"), "\nref", D.f(d_ref[0, i], "dxi"))
    Mr = R.f(ric_ref[0, i], "M"); print("|M|", np.linalg.norm(Mr), "|m|", np.linalg.norm(R.f(ric_ref[0, i], "m")), "|mt|", np.linalg.norm(R.f(ric_ref[0,i],"mt")), "|dx|", np.linalg.norm(D.f(d_ref[0,i],"dx")))
    ctx.close()
