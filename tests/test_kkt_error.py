"""KKT error at the evalKKT boundary (first piece of the on-device Newton loop):
OCPSolver::KKTError() without its STO term (reference src/solver/ocp_solver.cpp:429-431,
split_kkt_residual.hxx:90-104, contact_dynamics_data.hpp:204-206, constraint_component_data.hpp:122-124).
CPU: the oracle against numpy.  GPU: the HIP path against the oracle (1e-13: different summation order)."""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import (BUF_CDD, BUF_CON, BUF_CONE, BUF_KKT, GRID_IMPACT, GRID_TERMINAL, Records,
                               joint_limit_rows)

MC, CD = 4, 3


def _numpy_kkt_error(L, grids, kkt, cdd, con, rows):
    K, C, N = Records(L, "kkt"), Records(L, "cdd"), Records(L, "con")
    d = L.dims
    out = np.zeros(kkt.shape[0])
    row0 = d.nc_max - 5 * MC
    for b in range(kkt.shape[0]):
        e = 0.0
        for i, g in enumerate(grids):
            e += (K.f(kkt[b, i], "lx") ** 2).sum()
            if g.type == GRID_TERMINAL:
                continue
            imp = g.type == GRID_IMPACT
            e += (K.f(kkt[b, i], "Fx") ** 2).sum()
            if not imp:
                e += (K.f(kkt[b, i], "lu") ** 2).sum() + (K.f(kkt[b, i], "Pres")[:g.dims] ** 2).sum()
            e += (C.f(cdd[b, i], "la") ** 2).sum() + (C.f(cdd[b, i], "lf")[:g.dimf] ** 2).sum()
            e += (C.f(cdd[b, i], "IDC")[:d.nv + g.dimf] ** 2).sum()
            if not imp:
                e += (C.f(cdd[b, i], "lu_passive")[:d.np] ** 2).sum()
                act = np.array([g.time_stage >= r.level for r in rows])
                e += (N.f(con[b, i], "residual")[:len(rows)][act] ** 2).sum() + (N.f(con[b, i], "cmpl")[:len(rows)][act] ** 2).sum()
            n = 5 * (g.dimf // CD)
            e += (N.f(con[b, i], "residual")[row0:row0 + n] ** 2).sum() + (N.f(con[b, i], "cmpl")[row0:row0 + n] ** 2).sum()
        out[b] = np.sqrt(e)
    return out


def _data(L, grids, batch):
    kkt, cdd = pr.make_precondense_batch(L, grids, batch)
    con = pr.make_constraint_batch(L, grids, batch)
    rng = np.random.default_rng(2)
    Records(L, "con").f(con, "cmpl")[...] = 0.01 * rng.uniform(-1, 1, (batch, len(grids), L.dims.nc_max))
    return kkt, cdd, con


def test_oracle_kkt_error_matches_numpy(oracle):
    dims, grids, _ = pr.config_anymal_trot()
    L = oracle.layout(dims)
    kkt, cdd, con = _data(L, grids, 3)
    rows = joint_limit_rows(dims)
    ref = _numpy_kkt_error(L, grids, kkt, cdd, con, rows)
    got = oracle.kkt_error(L, grids, kkt, cdd, con, rows, MC, CD)
    assert np.allclose(got, ref, rtol=1e-13) and (ref > 1.0).all()
    # without the optional parts it is the SplitKKTResidual part alone: strictly smaller
    assert (oracle.kkt_error(L, grids, kkt) < got).all()


@pytest.mark.gpu
def test_gpu_kkt_error_matches_oracle(oracle):
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch = 5
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt, cdd, con = _data(L, grids, batch)
        rows = joint_limit_rows(dims)
        ctx.upload(BUF_KKT, kkt)
        assert np.allclose(ctx.kkt_error(), oracle.kkt_error(L, grids, kkt), rtol=1e-13)
        ctx.upload(BUF_CDD, cdd)
        ctx.set_constraint_rows(rows)
        ctx.set_friction_cones(MC, CD)
        ctx.upload(BUF_CON, con)
        ctx.upload(BUF_CONE, pr.make_cone_batch(L, grids, batch, MC))
        got = ctx.kkt_error()
        assert np.allclose(got, oracle.kkt_error(L, grids, kkt, cdd, con, rows, MC, CD), rtol=1e-13)
        assert np.array_equal(got, ctx.kkt_error())  # deterministic
    finally:
        ctx.close()
