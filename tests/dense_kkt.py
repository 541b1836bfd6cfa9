"""Dense assembly of the whole-horizon KKT system the Riccati recursion solves
(the horizon-level check the reference lacks: test/riccati/riccati_recursion_test.cpp:56-63
is an empty stub).  No switching-time (STO) variables: dts == 0.

Unknowns per stage i<N: dx_i (nx), du_i (nu, absent on impact grids), multiplier lam_i
(nx, = dlmdgmm_i), dxi_i (dims_i); terminal: dx_N, lam_N.
Equations (Newton-KKT system of the condensed OCP, cf. doc/ and riccati_factorizer.cpp):
  lam_0 row        : dx_0 = dx0_given
  dynamics i       : A_i dx_i + B_i du_i + Fx_i - dx_{i+1} = 0
  d/d dx_i         : Qxx dx + Qxu du + lx + A^T lam_{i+1} - lam_i + Phix^T dxi = 0
  d/d du_i         : Qxu^T dx + Quu du + lu + B^T lam_{i+1} + Phiu^T dxi = 0
  switching constr : Phix dx + Phiu du + P = 0
  terminal         : Qxx_N dx_N + lx_N - lam_N = 0
"""
import numpy as np

from robotoc_amd.types import GRID_IMPACT, Records


def solve_dense(L, grids, kkt, dx0):
    d = L.dims
    nv, nu, nx = d.nv, d.nu, 2 * d.nv
    K = Records(L, "kkt")
    N = len(grids) - 1
    # index maps
    idx = {}
    n = 0
    for i, g in enumerate(grids):
        idx[("x", i)] = (n, nx); n += nx
        idx[("l", i)] = (n, nx); n += nx
        if i < N and g.type != GRID_IMPACT:
            idx[("u", i)] = (n, nu); n += nu
        if i < N and g.dims > 0:
            idx[("xi", i)] = (n, g.dims); n += g.dims
    Mtx = np.zeros((n, n))
    rhs = np.zeros(n)

    def sl(key):
        o, m = idx[key]
        return slice(o, o + m)

    # rows are organised by "equation owner" = the unknown the stationarity is w.r.t.
    # lam_0 row: dx_0 = dx0
    Mtx[sl(("l", 0)), sl(("x", 0))] = np.eye(nx)
    rhs[sl(("l", 0))] = dx0
    for i, g in enumerate(grids):
        rec = kkt[i]
        Qxx = K.f(rec, "Qxx"); lx = K.f(rec, "lx")
        if i == N:
            r = sl(("x", i))
            Mtx[r, sl(("x", i))] = Qxx
            Mtx[r, sl(("l", i))] = -np.eye(nx)
            rhs[r] = -lx
            continue
        A = K.f(rec, "Fxx"); Fx = K.f(rec, "Fx")
        has_u = g.type != GRID_IMPACT
        B = np.zeros((nx, nu))
        if has_u:
            B[nv:, :] = K.f(rec, "Fvu")
        # dynamics row owned by lam_{i+1}
        r = sl(("l", i + 1))
        Mtx[r, sl(("x", i))] = A
        if has_u:
            Mtx[r, sl(("u", i))] = B
        Mtx[r, sl(("x", i + 1))] = -np.eye(nx)
        rhs[r] = -Fx
        # stationarity wrt dx_i
        r = sl(("x", i))
        Mtx[r, sl(("x", i))] = Qxx
        Mtx[r, sl(("l", i + 1))] = A.T
        Mtx[r, sl(("l", i))] = -np.eye(nx)
        rhs[r] = -lx
        if has_u:
            Qxu = K.f(rec, "Qxu"); Quu = K.f(rec, "Quu"); lu = K.f(rec, "lu")
            Mtx[r, sl(("u", i))] = Qxu
            ru = sl(("u", i))
            Mtx[ru, sl(("x", i))] = Qxu.T
            Mtx[ru, sl(("u", i))] = Quu
            Mtx[ru, sl(("l", i + 1))] = B.T
            rhs[ru] = -lu
        if g.dims > 0:
            m = g.dims
            Phix = K.f(rec, "Phix")[:m]; Phiu = K.f(rec, "Phiu")[:m]; P = K.f(rec, "Pres")[:m]
            rc = sl(("xi", i))
            Mtx[rc, sl(("x", i))] = Phix
            Mtx[rc, sl(("u", i))] = Phiu
            rhs[rc] = -P
            Mtx[sl(("x", i)), rc] = Phix.T
            Mtx[sl(("u", i)), rc] = Phiu.T
    sol = np.linalg.solve(Mtx, rhs)
    out = dict(dx=[], du=[], lam=[], dxi=[])
    for i, g in enumerate(grids):
        out["dx"].append(sol[sl(("x", i))])
        out["lam"].append(sol[sl(("l", i))])
        out["du"].append(sol[sl(("u", i))] if ("u", i) in idx else None)
        out["dxi"].append(sol[sl(("xi", i))] if ("xi", i) in idx else None)
    return out
