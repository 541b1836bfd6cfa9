"""UnconstrDynamics::condenseUnconstrDynamics / expandPrimal / expandDual (SURVEY 8a C6/C7,
reference src/dynamics/unconstr_dynamics.cpp:67-104) on the iiwa14 configuration.
CPU: the oracle against the closed form in numpy.  GPU: the HIP path through the C ABI against the
oracle (1e-13 relative: same sums, FMA contraction on the GPU only), then the full unconstrained
iteration condense -> backward -> forward -> expand."""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import BUF_CDD, BUF_DIR, BUF_DX0, BUF_KKT, Records


def _data(L, nstages, batch, seed=11):
    rng = np.random.default_rng(seed)
    nv = L.dims.nv
    kkt = Records(L, "kkt").zeros(batch, nstages)
    for b in range(batch):
        pr.fill_unconstr_instance(L, nstages, kkt[b], np.random.default_rng(seed + b))
    cdd = Records(L, "cdd").zeros(batch, nstages)
    C = Records(L, "cdd")
    C.f(cdd, "dIDCdqv")[...] = rng.uniform(-1, 1, (batch, nstages, nv, 2 * nv))
    C.f(cdd, "dIDda")[...] = rng.uniform(-1, 1, (batch, nstages, nv, nv))
    C.f(cdd, "IDC")[...] = rng.uniform(-1, 1, (batch, nstages, nv))
    C.f(cdd, "Qaa")[...] = rng.uniform(0.1, 1.0, (batch, nstages, nv))
    C.f(cdd, "la")[...] = rng.uniform(-1, 1, (batch, nstages, nv))
    # the full torque-cost Hessian for expandDual (unconstr_dynamics.cpp:99-104): symmetric, its diagonal = Qaa above
    Q = rng.uniform(-0.2, 0.2, (batch, nstages, nv, nv))
    Q = 0.5 * (Q + np.swapaxes(Q, -1, -2))
    idx = np.arange(nv)
    Q[..., idx, idx] = C.f(cdd, "Qaa")
    C.f(cdd, "MJtJinv")[...] = Q
    return kkt, cdd


def test_oracle_unconstr_condense_closed_form(oracle):
    dims, grids, meta = pr.config_iiwa14()
    L = oracle.layout(dims)
    n = len(grids)
    kkt, cdd = _data(L, n, 2)
    K, C = Records(L, "kkt"), Records(L, "cdd")
    nv = dims.nv
    ref = kkt.copy()
    for b in range(2):
        for i in range(n - 1):
            D = C.f(cdd[b, i], "dIDCdqv")
            dq, dv, da = D[:, :nv], D[:, nv:], C.f(cdd[b, i], "dIDda")
            w, ID, lut = C.f(cdd[b, i], "Qaa"), C.f(cdd[b, i], "IDC"), C.f(cdd[b, i], "la")
            luc = lut + w * ID
            Qxx, Qxu, Qaa = K.f(ref[b, i], "Qxx"), K.f(ref[b, i], "Qxu"), K.f(ref[b, i], "Quu")
            lx, la = K.f(ref[b, i], "lx"), K.f(ref[b, i], "lu")
            lx[:nv] += dq.T @ luc
            lx[nv:] += dv.T @ luc
            la += da.T @ luc
            W = np.diag(w)
            Qxx[:nv, :nv] += dq.T @ W @ dq
            Qxx[:nv, nv:] += dq.T @ W @ dv
            Qxx[nv:, :nv] = Qxx[:nv, nv:].T
            Qxx[nv:, nv:] += dv.T @ W @ dv
            Qaa += da.T @ W @ da
            Qxu[:nv] = (W @ dq).T @ da
            Qxu[nv:] = (W @ dv).T @ da
    oracle.unconstr_condense_batch(L, n, kkt, cdd)
    assert np.allclose(kkt, ref, rtol=1e-13, atol=1e-13)
    assert np.array_equal(kkt[:, -1], ref[:, -1])  # terminal stage untouched


def test_oracle_unconstr_expand_closed_form(oracle):
    dims, grids, meta = pr.config_iiwa14()
    L = oracle.layout(dims)
    n = len(grids)
    _, cdd = _data(L, n, 2)
    rng = np.random.default_rng(5)
    D, C = Records(L, "dir"), Records(L, "cdd")
    d = D.zeros(2, n)
    d[...] = rng.uniform(-1, 1, d.shape)
    ref = d.copy()
    nv, dt = dims.nv, meta["dt"]
    for b in range(2):
        for i in range(n - 1):
            J = C.f(cdd[b, i], "dIDCdqv")
            da = D.f(ref[b, i], "du").copy()
            dx = D.f(ref[b, i], "dx")
            du = C.f(cdd[b, i], "IDC") + J[:, :nv] @ dx[:nv] + J[:, nv:] @ dx[nv:] + C.f(cdd[b, i], "dIDda") @ da
            D.f(ref[b, i], "daf")[:nv] = da
            D.f(ref[b, i], "du")[:] = du
            D.f(ref[b, i], "dbetamu")[:nv] = (C.f(cdd[b, i], "la") + C.f(cdd[b, i], "MJtJinv") @ du) / dt  # FULL Quu
    oracle.unconstr_expand_batch(L, n, cdd, d, dt)
    assert np.allclose(d, ref, rtol=1e-13, atol=1e-13)


@pytest.mark.gpu
def test_gpu_unconstr_iteration_matches_oracle(oracle):
    from robotoc_amd import capi
    dims, grids, meta = pr.config_iiwa14()
    batch, n, dt = 6, len(grids), meta["dt"]
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt, cdd = _data(L, n, batch)
        dx0 = pr.make_dx0(L, batch)
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_CDD, cdd)
        ctx.upload(BUF_DX0, dx0)
        ctx.unconstr_condense()
        kkt_ref = kkt.copy()
        oracle.unconstr_condense_batch(L, n, kkt_ref, cdd)
        got = ctx.download_records(BUF_KKT, "kkt")
        assert np.allclose(got, kkt_ref, rtol=1e-13, atol=1e-13)
        # the rest of the iteration on the condensed system
        ctx.unconstr_backward(dt)
        ctx.unconstr_forward(dt)
        ctx.unconstr_expand(dt)
        assert (ctx.status() == 0).all()
        ric = Records(L, "ric").zeros(batch, n)
        d_ref = Records(L, "dir").zeros(batch, n)
        oracle.unconstr_sweep_batch(L, n, dt, kkt_ref.copy(), ric, d_ref, dx0=dx0)
        oracle.unconstr_expand_batch(L, n, cdd, d_ref, dt)
        d = ctx.download_records(BUF_DIR, "dir")
        D = Records(L, "dir")
        from helpers import rel_err
        for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu"):
            assert rel_err(D.f(d, f), D.f(d_ref, f)) <= 1e-9, f
    finally:
        ctx.close()
