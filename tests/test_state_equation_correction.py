"""Floating-base corrections of the linearised state equation:
correctLinearizeStateEquation / correctLinearizeImpactStateEquation / correctCostateDirection /
computeInitialStateDirection (reference src/dynamics/state_equation.cpp:68-109,
impact_state_equation.cpp:57-72).  CPU: the oracle against the closed form in numpy; GPU: the
HIP path (through the C ABI) against the oracle to 1e-13 relative (same 6-term dot products; the
GPU contracts them into FMAs, the C99 oracle does not)."""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import (BUF_DIR, BUF_DX0, BUF_KKT, BUF_SE3, GRID_IMPACT, GRID_TERMINAL, Records,
                               SE3_FQQ_INV, SE3_FQQ_PREV_INV, SE3_STRIDE)


def _data(L, grids, batch, seed=3):
    rng = np.random.default_rng(seed)
    kkt = pr.make_kkt_batch(L, grids, batch, mode="factory")
    d = Records(L, "dir").zeros(batch, len(grids))
    d[...] = rng.uniform(-1, 1, d.shape)
    dx0 = pr.make_dx0(L, batch)
    se3 = rng.uniform(-1, 1, (batch, len(grids), SE3_STRIDE))
    return kkt, d, dx0, se3


def test_oracle_matches_closed_form(oracle):
    dims, grids, _ = pr.config_anymal_trot()
    L = oracle.layout(dims)
    kkt, d, dx0, se3 = _data(L, grids, 2)
    K, D = Records(L, "kkt"), Records(L, "dir")
    nv = dims.nv
    kkt_ref, d_ref, dx0_ref = kkt.copy(), d.copy(), dx0.copy()
    for b in range(2):
        for i, g in enumerate(grids):
            inv = se3[b, i, SE3_FQQ_INV:SE3_FQQ_INV + 36].reshape(6, 6).T       # column-major
            pinv = se3[b, i, SE3_FQQ_PREV_INV:SE3_FQQ_PREV_INV + 36].reshape(6, 6).T
            dl = D.f(d_ref[b, i], "dlmdgmm")
            dl[:6] = -pinv.T @ dl[:6].copy()
            if g.type == GRID_TERMINAL:
                continue
            Fxx = K.f(kkt_ref[b, i], "Fxx")
            Fxx[:6, :6] = -inv @ Fxx[:6, :6].copy()
            Fx = K.f(kkt_ref[b, i], "Fx")
            Fx[:6] = -inv @ Fx[:6].copy()
            if g.type != GRID_IMPACT:
                Fxx[:6, nv:nv + 6] = -g.dt * inv
                fx = K.f(kkt_ref[b, i], "fx")
                fx[:6] = -inv @ fx[:6].copy()
        pinv0 = se3[b, 0, SE3_FQQ_PREV_INV:SE3_FQQ_PREV_INV + 36].reshape(6, 6).T
        dx0_ref[b, :6] = -pinv0 @ dx0[b, :6]
    oracle.state_correction_batch(L, grids, se3, kkt=kkt, dirs=d, dx0=dx0)
    assert np.allclose(kkt, kkt_ref, rtol=1e-14, atol=1e-14)
    assert np.allclose(d, d_ref, rtol=1e-14, atol=1e-14)
    assert np.allclose(dx0, dx0_ref, rtol=1e-14, atol=1e-14)
    # everything outside the 6x6 corners / 6-heads is untouched
    untouched = pr.make_kkt_batch(L, grids, 2, mode="factory")
    Fxx_new, Fxx_old = K.f(kkt, "Fxx"), K.f(untouched, "Fxx")
    assert np.array_equal(Fxx_new[..., 6:, :], Fxx_old[..., 6:, :])
    assert np.array_equal(K.f(kkt, "Qxx"), K.f(untouched, "Qxx"))


@pytest.mark.gpu
def test_gpu_state_corrections_match_oracle(oracle):
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch = 5
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt, d, dx0, se3 = _data(L, grids, batch)
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_DIR, d)
        ctx.upload(BUF_DX0, dx0)
        ctx.upload(BUF_SE3, se3)
        ctx.correct_state_equation()
        ctx.correct_costate_direction()
        ctx.compute_initial_state_direction()
        oracle.state_correction_batch(L, grids, se3, kkt=kkt, dirs=d, dx0=dx0)
        for got, ref in ((ctx.download_records(BUF_KKT, "kkt"), kkt),
                         (ctx.download_records(BUF_DIR, "dir"), d), (ctx.download(BUF_DX0, dx0.shape), dx0)):
            assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)
            # untouched entries are bitwise untouched
            assert (got != ref).sum() <= batch * len(grids) * 90
    finally:
        ctx.close()


@pytest.mark.gpu
def test_fixed_base_rejects_state_corrections():
    from robotoc_amd import capi
    dims, grids, _ = pr.config_iiwa14()
    ctx = capi.Context(dims, len(grids), 1, 0)
    try:
        ctx.set_grid(grids)
        with pytest.raises(capi.RtocError):
            ctx.correct_state_equation()  # hasFloatingBase() == false: nothing to correct
    finally:
        ctx.close()
