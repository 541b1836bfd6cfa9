"""The lane algebra of riccati_backward_rw_kernel (tests/rw_lane_model.py) against the CPU oracle: one regular, the lift, the
impact grid point and the grid point behind it, for both iCub shapes (nv = 32: NV = 2 tiles exactly; nv = 35: structured rows
shifted by 3 lanes / 3 rows).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from robotoc_amd import problems as pr
from robotoc_amd.types import GRID_IMPACT, Records

import rw_lane_model as m


@pytest.mark.parametrize("direct_ht", [False, True])
@pytest.mark.parametrize("nv", [32, 35])
def test_rw_lane_model_matches_oracle(nv, direct_ht):
    dims, grids, _ = pr.config_icub_jump(nv=nv)
    L = orc.layout(dims)
    NV, NU, NX = dims.nv, dims.nu, 2 * dims.nv
    c_ = m.Cfg(NV, NU)
    kkt = pr.make_kkt_batch_unique(L, grids, 1, seed=5)[0]
    Kr, Rr = Records(L, "kkt"), Records(L, "ric")
    ric = Rr.zeros(1, len(grids))[0]
    orc.riccati_backward(L, grids, kkt.copy(), ric)
    stages = [len(grids) - 2, 3]
    stages += [i for i, g in enumerate(grids) if g.type == GRID_IMPACT]
    stages += [i for i, g in enumerate(grids[:-1]) if g.type != GRID_IMPACT and g.dims == 0 and g.type != 0][:1]
    worst = 0.0
    for st in stages:
        g = grids[st]
        if g.type != GRID_IMPACT and g.dims > 0:
            continue   # switching-constraint grid points are the tile-split kernel's (one-stage launches)
        rec, nxt, out = kkt[st], ric[st + 1], ric[st]
        f = lambda n: Kr.f(rec, n).copy()   # noqa: E731
        pn, s, K, k = m.stage(c_, m.to_tiles(Rr.f(nxt, "P").copy(), NX), Rr.f(nxt, "s").copy(), f("Fxx"), f("Fvu"), f("Qxx"), f("Qxu"),
                              f("Quu"), f("Fx"), f("lx"), f("lu"), g.type == GRID_IMPACT, direct_ht=direct_ht)
        P = m.from_tiles(pn, NX)
        errs = {"P": np.abs(P - Rr.f(out, "P")).max() / np.abs(Rr.f(out, "P")).max(),
                "s": np.abs(s - Rr.f(out, "s")).max() / np.abs(Rr.f(out, "s")).max(), "asym": np.abs(P - P.T).max()}
        if g.type != GRID_IMPACT:
            errs["K"] = np.abs(K.T - Rr.f(out, "K")).max() / np.abs(Rr.f(out, "K")).max()
            errs["k"] = np.abs(k - Rr.f(out, "k")).max() / np.abs(Rr.f(out, "k")).max()
        print("nv", nv, "stage", st, "type", g.type, "mfma", m.stage.mfma, {n: float("%.1e" % v) for n, v in errs.items()})
        worst = max(worst, max(errs.values()))
    assert worst < 1e-10
