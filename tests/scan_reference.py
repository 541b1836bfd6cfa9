"""numpy statement of the horizon scan behind RTOC_OPT_BACKWARD_SCAN (test infrastructure).

The backward Riccati recursion (reference src/riccati/riccati_recursion.cpp:32-80) is a chain over
the grid points.  The scan replaces the chain by an associative combination of *interval elements*
(A, b, C, eta, J) -- the conditional value function of an interval [i, j) in its dual form

    V_{i->j}(x_i, x_j) = max_lam  1/2 x_i^T J x_i - eta^T x_i - 1/2 lam^T C lam - lam^T (x_j - A x_i - b)

(Saerkkae & Garcia-Fernandez, "Temporal parallelization of dynamic programming and linear quadratic
control", IEEE TAC 2023) -- so that P_i = J and s_i = eta of the suffix [i, N] come out of
ceil(log2(N+1)) combination levels.  A grid point with a switching constraint
(riccati_factorizer.cpp:58-89) is an element with the constraint eliminated inside it (projected
control Hessian), an impact grid point (:178-197) an element without control.

The functions below use exactly the factorised forms of the HIP kernels
(robotoc_amd/csrc/riccati_scan.hpp) and are checked against the serial oracle in
tests/test_scan_reference.py.
"""
import numpy as np

from robotoc_amd.types import GRID_IMPACT, GRID_TERMINAL, Records


def stage_element(L, g, rec):
    """Element of one grid point from its condensed KKT record.  Returns (A, b, C, eta, J, closed)."""
    K = Records(L, "kkt")
    nv, nu, nx = L.dims.nv, L.dims.nu, 2 * L.dims.nv
    Q = K.f(rec, "Qxx").copy()
    lx = K.f(rec, "lx").copy()
    if g.type == GRID_TERMINAL:
        z = np.zeros((nx, nx))
        return z, np.zeros(nx), z.copy(), -lx, Q, True
    A = K.f(rec, "Fxx").copy()
    b = K.f(rec, "Fx").copy()
    if g.type == GRID_IMPACT:
        return A, b, np.zeros((nx, nx)), -lx, Q, False
    S = K.f(rec, "Qxu")
    R = K.f(rec, "Quu")
    Bv = K.f(rec, "Fvu")
    lu = K.f(rec, "lu")
    Lc = np.linalg.cholesky(R)
    fs = lambda X: np.linalg.solve(Lc, X)  # forward substitution L^-1 X
    Zs, zl, Zb = fs(S.T), fs(lu), fs(Bv.T)
    ns = g.dims
    if ns > 0:
        Phix = K.f(rec, "Phix")[:ns]
        Phiu = K.f(rec, "Phiu")[:ns]
        Pres = K.f(rec, "Pres")[:ns]
        Zd = fs(Phiu.T)                         # nu x ns
        Ls = np.linalg.cholesky(Zd.T @ Zd)      # S = Phiu R^-1 Phiu^T
        Qd = np.linalg.solve(Ls, Zd.T).T        # Zd Ls^-T: orthonormal columns
        proj = lambda Z: Z - Qd @ (Qd.T @ Z)
        Ys = proj(Zs) + Qd @ np.linalg.solve(Ls, Phix)
        yl = proj(zl) + Qd @ np.linalg.solve(Ls, Pres)
        Zbp = proj(Zb)
    else:
        Ys, yl, Zbp = Zs, zl, Zb
    A[nv:, :] -= Zb.T @ Ys
    b[nv:] -= Zb.T @ yl
    C = np.zeros((nx, nx))
    C[nv:, nv:] = Zbp.T @ Zbp
    J = Q - Zs.T @ Ys - Ys.T @ Zs + Ys.T @ Ys
    eta = -lx + Zs.T @ yl + Ys.T @ zl - Ys.T @ yl
    return A, b, C, eta, J, False


def combine(e1, e2):
    """Element of [i, k) from those of [i, j) and [j, k)."""
    A1, b1, C1, eta1, J1, _ = e1
    A2, b2, C2, eta2, J2, closed2 = e2
    nx = A1.shape[0]
    M = np.eye(nx) + C1 @ J2
    T = np.linalg.solve(M, np.column_stack([A1, b1 + C1 @ eta2, C1]))
    Ta, tb, Tc = T[:, :nx], T[:, nx], T[:, nx + 1:]
    J = A1.T @ (J2 @ Ta) + J1
    J = 0.5 * (J + J.T)
    eta = A1.T @ (eta2 - J2 @ tb) + eta1
    if closed2:
        z = np.zeros((nx, nx))
        return z, np.zeros(nx), z.copy(), eta, J, True
    A = A2 @ Ta
    b = A2 @ tb + b2
    C = A2 @ Tc @ A2.T + C2
    C = 0.5 * (C + C.T)
    return A, b, C, eta, J, False


def scan_backward(L, grids, kkt):
    """Hillis-Steele suffix scan: returns (P, s) of every grid point, [stages, nx, nx] / [stages, nx]."""
    n = len(grids)
    e = [stage_element(L, g, kkt[i]) for i, g in enumerate(grids)]
    d = 1
    levels = 0
    while not all(x[5] for x in e):
        new = list(e)
        for i in range(n):
            if not e[i][5] and i + d < n:
                new[i] = combine(e[i], e[i + d])
        e = new
        d *= 2
        levels += 1
    P = np.stack([x[4] for x in e])
    s = np.stack([x[3] for x in e])
    return P, s, levels
