"""SplitSolution::integrate, Euclidean members (SURVEY 8f-2 / Appendix B k_expand_update; reference
src/core/split_solution.cpp:58-90).  CPU: the oracle against numpy.  GPU: through the C ABI against the
oracle, bit for bit (one multiply-add per entry; both sides may or may not contract it: 1e-15)."""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import BUF_DIR, BUF_SOL, BUF_STEP, GRID_IMPACT, Records


def _data(L, grids, batch):
    rng = np.random.default_rng(17)
    S, D = Records(L, "sol"), Records(L, "dir")
    sol = S.zeros(batch, len(grids))
    d = D.zeros(batch, len(grids))
    sol[...] = rng.uniform(-1, 1, sol.shape)
    d[...] = rng.uniform(-1, 1, d.shape)
    steps = np.stack([rng.uniform(0.2, 1.0, batch), rng.uniform(0.2, 1.0, batch)], axis=1)
    if L.dims.np == 6:  # a configuration on the manifold: unit quaternion of the free-flyer base
        o = L.sol.off[0]
        sol[..., o + 3:o + 7] /= np.linalg.norm(sol[..., o + 3:o + 7], axis=-1, keepdims=True)
    return sol, d, steps


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_oracle_integrate_closed_form(oracle):
    dims, grids, _ = pr.config_anymal_trot()
    L = oracle.layout(dims)
    sol, d, steps = _data(L, grids, 2)
    S, D = Records(L, "sol"), Records(L, "dir")
    nv, nu = dims.nv, dims.nu
    ref = sol.copy()
    for b in range(2):
        a = steps[b, 0]
        for i, g in enumerate(grids):
            s_, d_ = ref[b, i], d[b, i]
            imp = g.type == GRID_IMPACT
            S.f(s_, "q")[7:] += a * D.f(d_, "dx")[6:nv]
            S.f(s_, "v")[:] += a * D.f(d_, "dx")[nv:]
            S.f(s_, "a")[:] += a * D.f(d_, "daf")[:nv]
            if imp:
                S.f(s_, "u")[:] = 0.0
            else:
                S.f(s_, "u")[:] += a * D.f(d_, "du")
                S.f(s_, "nu_passive")[:6] += a * D.f(d_, "dnu_passive")[:6]
                S.f(s_, "xi")[:g.dims] += a * D.f(d_, "dxi")[:g.dims]
            S.f(s_, "lmd")[:] += a * D.f(d_, "dlmdgmm")[:nv]
            S.f(s_, "gmm")[:] += a * D.f(d_, "dlmdgmm")[nv:]
            S.f(s_, "beta")[:] += a * D.f(d_, "dbetamu")[:nv]
            S.f(s_, "f")[:g.dimf] += a * D.f(d_, "daf")[nv:nv + g.dimf]
            S.f(s_, "mu")[:g.dimf] += a * D.f(d_, "dbetamu")[nv:nv + g.dimf]
    o = L.sol.off[0]
    base = sol[:, :, o:o + 7].copy()
    oracle.integrate_solution_batch(L, grids, steps, d, sol)
    new_base = sol[:, :, o:o + 7].copy()
    sol[:, :, o:o + 7] = base
    assert np.allclose(sol, ref, rtol=1e-15, atol=1e-15)  # everything but the base
    # the free-flyer base: M <- M exp6(step dq[:6]) as 4x4 matrices (the SE(3) exponential restated independently: scipy-free series)
    for b in range(2):
        for i in range(len(grids)):
            xi = steps[b, 0] * D.f(d[b, i], "dx")[:6]
            E, pe = oracle.rbd_exp6(xi)
            R0 = _quat_R(base[b, i, 3:7])
            assert np.abs(_quat_R(new_base[b, i, 3:7]) - R0 @ E).max() < 1e-14
            assert np.abs(new_base[b, i, :3] - (base[b, i, :3] + R0 @ pe)).max() < 1e-14
            assert abs(np.linalg.norm(new_base[b, i, 3:7]) - 1.0) < 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["anymal", "iiwa14"])
def test_gpu_integrate_solution_matches_oracle(oracle, cfg):
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot() if cfg == "anymal" else pr.config_iiwa14()
    batch = 4
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        sol, d, steps = _data(L, grids, batch)
        ctx.upload(BUF_SOL, sol)
        ctx.upload(BUF_DIR, d)
        ctx.upload(BUF_STEP, steps)
        ctx.integrate_solution()
        got = ctx.download_records(BUF_SOL, "sol")
        oracle.integrate_solution_batch(L, grids, steps, d, sol)
        assert np.allclose(got, sol, rtol=1e-15, atol=1e-15)
    finally:
        ctx.close()
