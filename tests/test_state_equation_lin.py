"""rtoc_linearize_state_equation (state_equation_lin.hpp) against the restated reference lines (src/dynamics/
state_equation.cpp:8-66, impact_state_equation.cpp:8-57, the Fqq_inv / Fqq_prev_inv of :69-90 and computeInitialStateDirection
:99-109).  The SE(3) difference comes from the CPU restatement of pinocchio::difference (log6 of the relative placement,
checked against the exponential in tests/test_rigid_body.py); its Jacobians -- analytical on the device, forward mode
through the log -- are compared with central differences of that restatement on the manifold.  Parity of the SE(3) parts with
Pinocchio itself is unpinned (Pinocchio absent)."""
import numpy as np
import pytest

from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_CDD, BUF_DX0, BUF_KKT, BUF_SE3, BUF_SOL, GRID_IMPACT, Records

KKT_SCAL_H = 2


def _unit_quats(sol, o):
    sol[..., o + 3:o + 7] /= np.linalg.norm(sol[..., o + 3:o + 7], axis=-1, keepdims=True)


def _fd_jac(fun, q7, eps=1e-6):
    """d fun(q7 (+) e_k) / d e_k by central differences on SE(3)"""
    from oracle import oracle as orc
    J = np.zeros((6, 6))
    for k in range(6):
        e = np.zeros(6)
        e[k] = 1.0
        J[:, k] = (fun(orc.se3_integrate(q7, e, eps)) - fun(orc.se3_integrate(q7, e, -eps))) / (2 * eps)
    return J


@pytest.mark.gpu
def test_state_equation_linearisation_floating_base(oracle):
    dims, grids, _ = pr.config_anymal_trot()
    batch, n, nv = 3, len(grids), dims.nv
    ctx = capi.Context(dims, n, batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    rng = np.random.default_rng(12)
    S, K, C = Records(L, "sol"), Records(L, "kkt"), Records(L, "cdd")
    sol = rng.uniform(-1, 1, ctx.shape("sol"))
    # consecutive configurations stay within a fraction of a radian of each other, like iterates of a trajectory
    base = rng.uniform(-1, 1, (batch, 1, 7))
    sol[..., L.sol.off[0]:L.sol.off[0] + 7] = base + 0.3 * rng.uniform(-1, 1, (batch, n, 7))
    _unit_quats(sol, L.sol.off[0])
    x0 = np.concatenate([sol[:, 0, L.sol.off[0]:L.sol.off[0] + nv + 1] + 0.2 * rng.uniform(-1, 1, (batch, nv + 1)), rng.uniform(-1, 1, (batch, nv))], axis=1)
    x0[:, 3:7] /= np.linalg.norm(x0[:, 3:7], axis=-1, keepdims=True)
    kkt0, cdd0 = rng.uniform(-1, 1, ctx.shape("kkt")), rng.uniform(-1, 1, ctx.shape("cdd"))
    ctx.upload(BUF_SOL, sol)
    ctx.upload(BUF_KKT, kkt0)
    ctx.upload(BUF_CDD, cdd0)
    ctx.set_initial_state(x0)
    ctx.linearize_state_equation()
    ctx.sync()
    kkt, cdd = ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd")
    se3 = ctx.download(BUF_SE3, (batch, n, 72))
    dx0 = ctx.download(BUF_DX0, (batch, 2 * nv))
    worst = dict(val=0.0, jac=0.0)
    for b in range(batch):
        for i in range(n - 1):
            g = grids[i]
            imp = g.type == GRID_IMPACT
            dt = 0.0 if imp else g.dt
            s, sn = sol[b, i], sol[b, i + 1]
            q, v, a, lmd, gmm = S.f(s, "q"), S.f(s, "v"), S.f(s, "a"), S.f(s, "lmd"), S.f(s, "gmm")
            qn, vn, lmdn, gmmn = S.f(sn, "q"), S.f(sn, "v"), S.f(sn, "lmd"), S.f(sn, "gmm")
            qp = S.f(sol[b, i - 1], "q") if i > 0 else x0[b, :nv + 1]
            # residuals (state_equation.cpp:16-19 / impact :14-15)
            Fq = np.concatenate([oracle.se3_difference(qn[:7], q[:7]), q[7:] - qn[7:]]) + dt * v
            Fv = v + (a if imp else dt * a) - vn
            got = K.f(kkt[b, i], "Fx")
            worst["val"] = max(worst["val"], np.abs(got - np.concatenate([Fq, Fv])).max())
            # Jacobians by central differences of the restated difference
            Fqq = _fd_jac(lambda x: oracle.se3_difference(qn[:7], x), q[:7])
            Fqq_prev = _fd_jac(lambda x: oracle.se3_difference(x, qp[:7]), q[:7])
            Jn = _fd_jac(lambda x: oracle.se3_difference(x, q[:7]), qn[:7])   # dSubtractConfiguration_dq0(q, q_next)
            Fxx = K.f(kkt[b, i], "Fxx")
            top = np.zeros((nv, 2 * nv))
            top[:, :nv] = np.eye(nv)
            top[:6, :6] = Fqq
            top[:, nv:] = dt * np.eye(nv)
            worst["jac"] = max(worst["jac"], np.abs(Fxx[:nv] - top).max())
            assert np.array_equal(Fxx[nv:], K.f(kkt0[b, i], "Fxx")[nv:])  # the bottom half belongs to the condensation
            worst["jac"] = max(worst["jac"], np.abs(se3[b, i, :36].reshape(6, 6).T - np.linalg.inv(Jn)).max(),
                               np.abs(se3[b, i, 36:].reshape(6, 6).T - np.linalg.inv(Fqq_prev)).max())
            # multiplier terms (:41-56 / impact :40-54), added to what was there
            lx = K.f(kkt0[b, i], "lx").copy()
            lx[:6] += Fqq.T @ lmdn[:6] + Fqq_prev.T @ lmd[:6]
            lx[6:nv] += lmdn[6:] - lmd[6:]
            lx[nv:] += dt * lmdn + gmmn - gmm
            worst["jac"] = max(worst["jac"], np.abs(K.f(kkt[b, i], "lx") - lx).max())
            assert np.allclose(C.f(cdd[b, i], "la"), C.f(cdd0[b, i], "la") + (gmmn if imp else dt * gmmn), atol=1e-14)
            if not imp:  # STO sensitivities (:58-63)
                assert np.allclose(K.f(kkt[b, i], "hx")[nv:], K.f(kkt0[b, i], "hx")[nv:] + lmdn, atol=1e-14)
                assert np.allclose(C.f(cdd[b, i], "ha"), C.f(cdd0[b, i], "ha") + gmmn, atol=1e-14)
                assert np.allclose(K.f(kkt[b, i], "fx"), np.concatenate([v, a]), atol=1e-15)
                assert abs(K.f(kkt[b, i], "scal")[KKT_SCAL_H] - (K.f(kkt0[b, i], "scal")[KKT_SCAL_H] + lmdn @ v + gmmn @ a)) < 1e-13
            if i == 0:  # computeInitialStateDirection (:99-109)
                d0 = np.concatenate([oracle.se3_difference(q[:7], x0[b, :7]), x0[b, 7:nv + 1] - q[7:]])
                d0[:6] = -np.linalg.inv(Fqq_prev) @ d0[:6]
                worst["jac"] = max(worst["jac"], np.abs(dx0[b, :nv] - d0).max())
                assert np.allclose(dx0[b, nv:], x0[b, nv + 1:] - v, atol=1e-15)
        # terminal grid point (terminal_state_equation.cpp:8-28)
        sT, qpT = sol[b, n - 1], S.f(sol[b, n - 2], "q")
        qT, lmdT, gmmT = S.f(sT, "q"), S.f(sT, "lmd"), S.f(sT, "gmm")
        FqqT = _fd_jac(lambda x: oracle.se3_difference(x, qpT[:7]), qT[:7])
        lx = K.f(kkt0[b, n - 1], "lx").copy()
        lx[:6] += FqqT.T @ lmdT[:6]
        lx[6:nv] -= lmdT[6:]
        lx[nv:] -= gmmT
        worst["jac"] = max(worst["jac"], np.abs(K.f(kkt[b, n - 1], "lx") - lx).max(),
                           np.abs(se3[b, n - 1, 36:].reshape(6, 6).T - np.linalg.inv(FqqT)).max())
    print("worst deviation: values %.2e, Jacobian-dependent %.2e" % (worst["val"], worst["jac"]))
    assert worst["val"] < 1e-13 and worst["jac"] < 2e-7
    ctx.close()


@pytest.mark.gpu
def test_state_equation_linearisation_fixed_base():
    dims, grids, meta = pr.config_iiwa14()
    batch, n, nv = 2, len(grids), dims.nv
    ctx = capi.Context(dims, n, batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    rng = np.random.default_rng(13)
    S, K = Records(L, "sol"), Records(L, "kkt")
    sol = rng.uniform(-1, 1, ctx.shape("sol"))
    kkt0 = rng.uniform(-1, 1, ctx.shape("kkt"))
    x0 = rng.uniform(-1, 1, (batch, 2 * nv))
    ctx.upload(BUF_SOL, sol)
    ctx.upload(BUF_KKT, kkt0)
    ctx.set_initial_state(x0)
    ctx.linearize_state_equation()
    kkt = ctx.download_records(BUF_KKT, "kkt")
    dx0 = ctx.download(BUF_DX0, (batch, 2 * nv))
    for b in range(batch):
        for i in range(n - 1):
            dt = grids[i].dt
            s, sn = sol[b, i], sol[b, i + 1]
            q, v, a = S.f(s, "q")[:nv], S.f(s, "v"), S.f(s, "a")
            assert np.allclose(K.f(kkt[b, i], "Fx"), np.concatenate([q + dt * v - S.f(sn, "q")[:nv], v + dt * a - S.f(sn, "v")]), atol=1e-15)
            top = np.hstack([np.eye(nv), dt * np.eye(nv)])
            assert np.array_equal(K.f(kkt[b, i], "Fxx")[:nv], top)
            lx = K.f(kkt0[b, i], "lx").copy()
            lx[:nv] += S.f(sn, "lmd") - S.f(s, "lmd")
            lx[nv:] += dt * S.f(sn, "lmd") + S.f(sn, "gmm") - S.f(s, "gmm")
            assert np.allclose(K.f(kkt[b, i], "lx"), lx, atol=1e-14)
        assert np.allclose(dx0[b], x0[b] - np.concatenate([S.f(sol[b, 0], "q")[:nv], S.f(sol[b, 0], "v")]), atol=1e-15)
    ctx.close()
