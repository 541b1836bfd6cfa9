"""Pins the inequality-row algebra to the reference's OWN sources (oracle/_ref/librtoc_ref.so: robotoc's Constraints object
with the six joint-limit components, FrictionCone / ImpactFrictionCone, ContactWrenchCone and pdipm.hxx, compiled from
/root/reference against the Eigen stand-in): (a) the numpy restatement of their EVALUATION (tests/constraint_restatement.py,
which the GPU tests hold the device kernels to), (b) the C oracle's CONDENSATION / EXPANSION of these rows
(oracle/rtoc_oracle_condense.c: what rtoc_condense / rtoc_expand are held to).  The frame kinematics the cones read are
injected (random rotations and Jacobians): what is pinned is the reference's composition of them."""
import ctypes as C

import numpy as np
import pytest

import constraint_restatement as cr
from oracle import ref
from robotoc_amd.grid import Grid
from robotoc_amd.types import GRID_IMPACT, GRID_INTERMEDIATE, Records, anymal_dims, cone_dgdf_off, joint_limit_rows

pytestmark = pytest.mark.skipif(not ref.available(), reason="needs /root/reference (oracle/_ref)")


def _rot(rng):
    q = rng.normal(size=4)
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("time_stage,impact,active", [(0, False, 0b1111), (1, False, 0b1001), (5, False, 0b0110), (3, True, 0b0110), (7, False, 0)])
def test_joint_limits_and_friction_cones_against_the_reference_sources(oracle, time_stage, impact, active):
    dims = anymal_dims()
    nv, nu, nc, nx = dims.nv, dims.nu, 4, 2 * dims.nv
    rng = np.random.default_rng(100 + time_stage)
    rows = joint_limit_rows(dims)
    q = rng.uniform(-1, 1, nv + 1)
    v, u = rng.uniform(-2, 2, nv), rng.uniform(-30, 30, nu)
    f = np.zeros((nc, 6))
    f[:, :3] = rng.uniform(-20, 20, (nc, 3))
    f[:, 2] = rng.uniform(10, 80, nc)
    mu = rng.uniform(0.4, 0.9, nc)
    Rs = np.array([_rot(rng) for _ in range(nc)])
    Rwf = np.array([_rot(rng) for _ in range(nc)])
    Jw = rng.uniform(-1, 1, (nc, 6, nv))      # world-aligned frame Jacobians (the angular rows are what the cone reads)
    Jl = np.array([np.vstack([Rwf[c].T @ Jw[c, :3], Rwf[c].T @ Jw[c, 3:]]) for c in range(nc)])   # pinocchio LOCAL
    qmin, qmax, vmax, umax = -rng.uniform(0.5, 1.5, nu), rng.uniform(0.5, 1.5, nu), rng.uniform(1, 3, nu), rng.uniform(20, 40, nu)
    bounds = np.concatenate([-qmin, qmax, vmax, vmax, umax, umax])
    barrier, tau = 1.0e-3, 0.995
    act = [c for c in range(nc) if (active >> c) & 1]
    dimf = 3 * len(act)
    nrow = 6 * nu + 5 * nc
    L = ref.lib()
    L.ref_constraints_stage.restype = C.c_int

    def call(phase, slack, dual, residual, cmpl, lx, lu, lf, Qxx, Quu, Qqf, Qff, dx=None, du=None, df=None):
        cond, dslack, ddual = np.zeros(nrow), np.zeros(nrow), np.zeros(nrow)
        steps, dgdq, dgdf = np.ones(2), np.zeros((nc, nv, 5)), np.zeros((nc, 3, 5))
        z = np.zeros(max(nx, 1))
        rc = L.ref_constraints_stage(nv, nu, nc, 3, time_stage, int(impact), C.c_uint(active), _d(mu), _d(np.ascontiguousarray(Rs)),
                                     _d(np.ascontiguousarray(Rwf)), _d(np.ascontiguousarray(Jl.transpose(0, 2, 1))),
                                     _d(np.concatenate([qmin, qmax, vmax, umax])), 1, C.c_double(0), C.c_double(0), C.c_double(barrier),
                                     C.c_double(tau), _d(q), _d(v), _d(u), _d(np.ascontiguousarray(f)), phase, _d(slack), _d(dual),
                                     _d(residual), _d(cmpl), _d(cond), _d(dslack), _d(ddual), _d(lx), _d(lu), _d(lf), _d(Qxx), _d(Quu),
                                     _d(Qqf), _d(Qff), _d(dx if dx is not None else z), _d(du if du is not None else z),
                                     _d(df if df is not None else z), _d(steps), _d(dgdq), _d(dgdf))
        assert rc == 0
        return cond, dslack, ddual, steps, dgdq.transpose(0, 2, 1), dgdf.transpose(0, 2, 1)

    # ---- the restatement: values of all rows (cone rows by contact index) ----
    jact = cr.joint_limit_active(rows, time_stage, impact)
    g_joint = cr.joint_limit_values(rows, bounds, q, v, u, floating=True)
    cones = {c: cr.friction_cone_rows(mu[c], Rs[c], Rwf[c], Jw[c, 3:], f[c], exact_jacobian=False) for c in act}
    # (a1) setSlackAndDual
    slack, dual = np.zeros(nrow), np.zeros(nrow)
    zeros = lambda *s: np.zeros(s)
    call(1, slack, dual, zeros(nrow), zeros(nrow), zeros(nx), zeros(nu), zeros(max(dimf, 1)), zeros(nx, nx), zeros(nu, nu),
         zeros(max(dimf, 1), nv), zeros(max(dimf, 1), max(dimf, 1)))
    s_j, d_j = cr.init_slack_dual(g_joint, barrier)
    assert np.allclose(slack[:6 * nu][jact], s_j[jact], rtol=1e-14, atol=0) and np.allclose(dual[:6 * nu][jact], d_j[jact], rtol=1e-14, atol=0)
    for c in act:
        s_c, d_c = cr.init_slack_dual(cones[c][0], barrier)
        rr = slice(6 * nu + 5 * c, 6 * nu + 5 * c + 5)
        assert np.allclose(slack[rr], s_c, rtol=1e-13, atol=1e-13) and np.allclose(dual[rr], d_c, rtol=1e-13, atol=1e-13)
    # (a2) linearizeConstraints at random positive slack / dual, then (b) condense and expand in the same call
    slack, dual = rng.uniform(0.1, 2.0, nrow), rng.uniform(0.1, 2.0, nrow)
    residual, cmpl = np.zeros(nrow), np.zeros(nrow)
    lx0, lu0, lf0 = rng.uniform(-1, 1, nx), rng.uniform(-1, 1, nu), rng.uniform(-1, 1, max(dimf, 1))
    A = rng.uniform(-1, 1, (nx + nu + max(dimf, 1),) * 2)
    H = A @ A.T
    Qxx0, Quu0 = np.asfortranarray(H[:nx, :nx]), np.asfortranarray(H[nx:nx + nu, nx:nx + nu])
    Qqf0, Qff0 = np.asfortranarray(H[:nv, nx + nu:nx + nu + dimf]), np.asfortranarray(H[nx + nu:nx + nu + dimf, nx + nu:nx + nu + dimf])
    dx, du, df = rng.uniform(-1, 1, nx), rng.uniform(-1, 1, nu), rng.uniform(-1, 1, max(dimf, 1))
    lx, lu, lf = lx0.copy(), lu0.copy(), lf0.copy()
    Qxx, Quu = Qxx0.copy(order="F"), Quu0.copy(order="F")
    Qqf = np.asfortranarray(Qqf0.copy()) if dimf else zeros(1, nv)
    Qff = np.asfortranarray(Qff0.copy()) if dimf else zeros(1, 1)
    s2, d2 = slack.copy(), dual.copy()
    cond, dslack, ddual, steps, dgdq, dgdf = call(2 | 4 | 8, s2, d2, residual, cmpl, lx, lu, lf, Qxx, Quu, Qqf, Qff, dx, du, df)
    # evaluation against the restatement
    worst = dict(residual=0.0, cmpl=0.0, dgdq=0.0, dgdf=0.0)
    for r in np.nonzero(jact)[0]:
        worst["residual"] = max(worst["residual"], abs(residual[r] - (g_joint[r] + slack[r])))
        worst["cmpl"] = max(worst["cmpl"], abs(cmpl[r] - (slack[r] * dual[r] - barrier)))
    for c in act:
        gc, dq_c, df_c = cones[c]
        rr = slice(6 * nu + 5 * c, 6 * nu + 5 * c + 5)
        worst["residual"] = max(worst["residual"], np.abs(residual[rr] - (gc + slack[rr])).max())
        worst["cmpl"] = max(worst["cmpl"], np.abs(cmpl[rr] - (slack[rr] * dual[rr] - barrier)).max())
        worst["dgdq"] = max(worst["dgdq"], np.abs(dgdq[c] - dq_c).max())
        worst["dgdf"] = max(worst["dgdf"], np.abs(dgdf[c] - df_c).max())
    print("evaluation vs the reference sources:", {k: "%.1e" % e for k, e in worst.items()})
    assert max(worst.values()) < 1e-12
    # ---- (b) the C oracle's condensation / expansion on records carrying the same data ----
    g = Grid(GRID_IMPACT if impact else GRID_INTERMEDIATE, 0, 0, 0, dimf, 0, 10, -1 if impact else time_stage, 0.0 if impact else 0.02)
    gt = Grid(3, 0, 0, 0, 0, 0, 0, 11, 0.0)
    Lo = oracle.layout(dims)
    K, D, N, R = Records(Lo, "kkt"), Records(Lo, "cdd"), Records(Lo, "con"), Records(Lo, "dir")
    kkt, cdd, con, dirs = K.zeros(1, 2), D.zeros(1, 2), N.zeros(1, 2), R.zeros(1, 2)
    # gradients after linearizeConstraints = before condensation: the start values plus the restatement's increments
    lxa, lua = cr.joint_limit_gradients(rows, jact, dual, nv, nu)
    lfa = np.zeros(max(dimf, 1))
    _off = cone_dgdf_off
    cone = np.zeros((1, 2, _off(nv, 4) + 64))
    for k, c in enumerate(act):
        rr = slice(6 * nu + 5 * c, 6 * nu + 5 * c + 5)
        lxa[:nv] += cones[c][1].T @ dual[rr]
        lfa[3 * k:3 * k + 3] += cones[c][2].T @ dual[rr]
        cone[0, 0, k * 5 * nv:(k + 1) * 5 * nv] = cones[c][1].T.reshape(-1)
        cone[0, 0, _off(nv, 4) + 15 * k:_off(nv, 4) + 15 * k + 15] = cones[c][2].T.reshape(-1)
    K.f(kkt[0, 0], "Qxx")[:] = Qxx0
    K.f(kkt[0, 0], "Quu")[:] = Quu0
    K.f(kkt[0, 0], "lx")[:] = lx0 + lxa
    K.f(kkt[0, 0], "lu")[:] = lu0 + lua
    if dimf:
        D.f(cdd[0, 0], "Qqf")[:, :dimf] = Qqf0
        D.f(cdd[0, 0], "Qff")[:dimf, :dimf] = Qff0
        D.f(cdd[0, 0], "lf")[:dimf] = lf0[:dimf] + lfa[:dimf]
    nc_max, row0 = dims.nc_max, dims.nc_max - 20
    for name, arr in (("slack", slack), ("dual", dual), ("residual", residual), ("cmpl", cmpl)):
        N.f(con[0, 0], name)[:6 * nu] = arr[:6 * nu]
        for k, c in enumerate(act):   # the records compact the cone rows by ACTIVE contact
            N.f(con[0, 0], name)[row0 + 5 * k:row0 + 5 * k + 5] = arr[6 * nu + 5 * c:6 * nu + 5 * c + 5]
    grids = [g, gt]
    oracle.pdipm_condense_batch(Lo, grids, rows, kkt, con)
    if dimf:
        oracle.cone_condense_batch(Lo, grids, 4, 3, cone, kkt, cdd, con)
    w2 = dict(Qxx=np.abs(K.f(kkt[0, 0], "Qxx") - Qxx).max(), Quu=np.abs(K.f(kkt[0, 0], "Quu") - Quu).max(),
              lx=np.abs(K.f(kkt[0, 0], "lx") - lx).max(), lu=np.abs(K.f(kkt[0, 0], "lu") - lu).max())
    if dimf:
        w2["Qqf"] = np.abs(D.f(cdd[0, 0], "Qqf")[:, :dimf] - Qqf).max()
        w2["Qff"] = np.abs(D.f(cdd[0, 0], "Qff")[:dimf, :dimf] - Qff).max()
        w2["lf"] = np.abs(D.f(cdd[0, 0], "lf")[:dimf] - lf[:dimf]).max()
    print("condensation, C oracle vs the reference sources:", {k: "%.1e" % e for k, e in w2.items()})
    assert max(w2.values()) < 1e-10
    # expansion and step sizes
    R.f(dirs[0, 0], "dx")[:] = dx
    R.f(dirs[0, 0], "du")[:] = du
    if dimf:
        R.f(dirs[0, 0], "daf")[nv:nv + dimf] = df[:dimf]
    st = oracle.pdipm_expand_batch(Lo, grids, rows, con, dirs, tau)
    if dimf:
        oracle.cone_expand_batch(Lo, grids, 4, 3, cone, con, dirs, tau, st)
    w3 = dict(dslack=0.0, ddual=0.0)
    for r in np.nonzero(jact)[0]:
        w3["dslack"] = max(w3["dslack"], abs(N.f(con[0, 0], "dslack")[r] - dslack[r]))
        w3["ddual"] = max(w3["ddual"], abs(N.f(con[0, 0], "ddual")[r] - ddual[r]))
    for k, c in enumerate(act):
        a, b = slice(row0 + 5 * k, row0 + 5 * k + 5), slice(6 * nu + 5 * c, 6 * nu + 5 * c + 5)
        w3["dslack"] = max(w3["dslack"], np.abs(N.f(con[0, 0], "dslack")[a] - dslack[b]).max())
        w3["ddual"] = max(w3["ddual"], np.abs(N.f(con[0, 0], "ddual")[a] - ddual[b]).max())
    w3["steps"] = float(np.abs(st[0] - steps).max())
    print("expansion and fraction-to-boundary steps, C oracle vs the reference sources:", {k: "%.1e" % e for k, e in w3.items()}, steps)
    assert max(w3.values()) < 1e-10


@pytest.mark.parametrize("active,impact", [(0b11, False), (0b10, False), (0b11, True), (0b01, True)])
def test_contact_wrench_cone_against_the_reference_sources(oracle, active, impact):
    """ContactWrenchCone (17 rows per active surface contact, src/constraints/contact_wrench_cone.cpp) and, on impact grids,
    ImpactWrenchCone (src/constraints/impact_wrench_cone.cpp: the same algebra on the impulse, impact level): the cone matrix of
    computeCone / updateCone, evaluation, condensation into Qff / lf, expansion and step sizes -- reference sources vs the C
    oracle (iCub: nv = 35, two soles)."""
    from robotoc_amd.types import icub_dims
    dims = icub_dims(35, nc_max=6 * 29 + 34 + 6)
    nv, nu, nc, nx = dims.nv, dims.nu, 2, 2 * dims.nv
    rng = np.random.default_rng(7 + active)
    rows = joint_limit_rows(dims)
    q, v, u = rng.uniform(-1, 1, nv + 1), rng.uniform(-2, 2, nv), rng.uniform(-30, 30, nu)
    f = rng.uniform(-5, 5, (nc, 6))
    f[:, 2] = rng.uniform(50, 150, nc)
    mu = np.array([0.6, 0.8])
    X, Y = 0.1, 0.05
    lim = np.concatenate([-np.full(nu, 2.0), np.full(nu, 2.0), np.full(nu, 5.0), np.full(nu, 60.0)])
    barrier, tau, time_stage = 1.0e-3, 0.995, 4
    act = [c for c in range(nc) if (active >> c) & 1]
    dimf = 6 * len(act)
    nrow = 6 * nu + 17 * nc
    L = ref.lib()
    eye = np.tile(np.eye(3).reshape(-1), nc)
    slack, dual = rng.uniform(0.1, 2.0, nrow), rng.uniform(0.1, 2.0, nrow)
    residual, cmpl, cond, dslack, ddual = (np.zeros(nrow) for _ in range(5))
    lx, lu, lf0 = np.zeros(nx), np.zeros(nu), rng.uniform(-1, 1, dimf)
    A = rng.uniform(-1, 1, (dimf, dimf))
    Qff0 = np.asfortranarray(A @ A.T)
    lf, Qff = lf0.copy(), Qff0.copy(order="F")
    Qxx, Quu, Qqf = np.zeros((nx, nx), order="F"), np.zeros((nu, nu), order="F"), np.zeros((dimf, nv)).T.copy(order="F")
    dx, du, df = rng.uniform(-1, 1, nx), rng.uniform(-1, 1, nu), rng.uniform(-1, 1, dimf)
    steps, dgdq, dgdf = np.ones(2), np.zeros(nc * 5 * nv), np.zeros(nc * 15)
    rc = L.ref_constraints_stage(nv, nu, nc, 6, time_stage, int(impact), C.c_uint(active), _d(mu), None, _d(eye), _d(np.zeros(nc * 6 * nv)), _d(lim), 2,
                                 C.c_double(X), C.c_double(Y), C.c_double(barrier), C.c_double(tau), _d(q), _d(v), _d(u),
                                 _d(np.ascontiguousarray(f)), 2 | 4 | 8, _d(slack), _d(dual), _d(residual), _d(cmpl), _d(cond), _d(dslack),
                                 _d(ddual), _d(lx), _d(lu), _d(lf), _d(Qxx), _d(Quu), _d(Qqf), _d(Qff), _d(dx), _d(du), _d(df), _d(steps),
                                 _d(dgdq), _d(dgdf))
    assert rc == 0
    worst = dict(residual=0.0, cmpl=0.0)
    cones = {}
    for c in act:
        cones[c] = oracle.wrench_cone_matrix(X, Y, mu[c])
        rr = slice(6 * nu + 17 * c, 6 * nu + 17 * c + 17)
        worst["residual"] = max(worst["residual"], np.abs(residual[rr] - (cones[c] @ f[c] + slack[rr])).max())
        worst["cmpl"] = max(worst["cmpl"], np.abs(cmpl[rr] - (slack[rr] * dual[rr] - barrier)).max())
    print("wrench cone evaluation vs the reference sources:", {k: "%.1e" % e for k, e in worst.items()})
    assert max(worst.values()) < 1e-12
    Lo = oracle.layout(dims)
    D, N, R = Records(Lo, "cdd"), Records(Lo, "con"), Records(Lo, "dir")
    cdd, con, dirs = D.zeros(1, 2), N.zeros(1, 2), R.zeros(1, 2)
    grids = [Grid(GRID_IMPACT, 0, 0, 0, dimf, 0, 10, -1, 0.0) if impact else Grid(GRID_INTERMEDIATE, 0, 0, 0, dimf, 0, 10, time_stage, 0.02),
             Grid(3, 0, 0, 0, 0, 0, 0, 11, 0.0)]
    cone = np.zeros((1, 2, 2 * 102 + 8))
    lfa = np.zeros(dimf)
    row0 = dims.nc_max - 34
    for k, c in enumerate(act):
        rr = slice(6 * nu + 17 * c, 6 * nu + 17 * c + 17)
        cone[0, 0, 102 * k:102 * k + 102] = cones[c].T.reshape(-1)
        lfa[6 * k:6 * k + 6] += cones[c].T @ dual[rr]      # evalDerivatives (:169-204)
        for name, arr in (("slack", slack), ("dual", dual), ("residual", residual), ("cmpl", cmpl)):
            N.f(con[0, 0], name)[row0 + 17 * k:row0 + 17 * k + 17] = arr[rr]
    D.f(cdd[0, 0], "Qff")[:dimf, :dimf] = Qff0
    D.f(cdd[0, 0], "lf")[:dimf] = lf0 + lfa
    oracle.wrench_condense_batch(Lo, grids, 2, cone, cdd, con)
    w2 = dict(Qff=np.abs(D.f(cdd[0, 0], "Qff")[:dimf, :dimf] - Qff).max(), lf=np.abs(D.f(cdd[0, 0], "lf")[:dimf] - lf).max())
    R.f(dirs[0, 0], "daf")[nv:nv + dimf] = df
    st = np.ones((1, 2))
    oracle.wrench_expand_batch(Lo, grids, 2, cone, con, dirs, tau, st)
    for k, c in enumerate(act):
        a, b = slice(row0 + 17 * k, row0 + 17 * k + 17), slice(6 * nu + 17 * c, 6 * nu + 17 * c + 17)
        w2["dslack"] = max(w2.get("dslack", 0.0), np.abs(N.f(con[0, 0], "dslack")[a] - dslack[b]).max())
        w2["ddual"] = max(w2.get("ddual", 0.0), np.abs(N.f(con[0, 0], "ddual")[a] - ddual[b]).max())
    # the reference's step sizes include the joint-limit rows; compare the cone part: re-run the reference on the joint rows alone
    print("wrench cone condensation / expansion, C oracle vs the reference sources:", {k: "%.1e" % e for k, e in w2.items()})
    assert max(w2.values()) < 1e-10


@pytest.mark.parametrize("time_stage,impact", [(0, False), (4, False), (2, True)])
def test_acceleration_limits_against_the_reference_sources(oracle, time_stage, impact):
    """JointAccelerationLowerLimit / JointAccelerationUpperLimit (src/constraints/joint_acceleration_{lower,upper}_limit.cpp) inside
    the reference's Constraints object vs (a) the restatement of their evaluation -- g = sign a - bound on the tail nu entries of
    a, la += sign dual -- and (b) the C oracle's RTOC_VAR_A rows: Qaa.diagonal() += dual / slack, la += sign cond ahead of the
    contact-dynamics condensation, dslack = -sign da - residual, fraction-to-boundary steps."""
    from robotoc_amd.types import VAR_A, BoxRow
    dims = anymal_dims()
    nv, nu, npv = dims.nv, dims.nu, dims.np
    rng = np.random.default_rng(300 + time_stage)
    rows = [BoxRow(VAR_A, npv + j, sign, 0) for sign in (-1, +1) for j in range(nu)]
    nrow = 2 * nu
    a = rng.uniform(-3, 3, nv)
    amin, amax = -rng.uniform(2, 5, nu), rng.uniform(2, 5, nu)
    bounds = np.concatenate([-amin, amax])
    barrier, tau = 1.0e-3, 0.995
    L = ref.lib()
    L.ref_accel_limits_stage.restype = C.c_int

    def call(phase, slack, dual, residual, cmpl, Qaa, la, da=None):
        cond, dslack, ddual, steps = np.zeros(nrow), np.zeros(nrow), np.zeros(nrow), np.ones(2)
        rc = L.ref_accel_limits_stage(nv, nu, time_stage, int(impact), _d(amin), _d(amax), C.c_double(barrier), C.c_double(tau), _d(a), phase,
                                      _d(slack), _d(dual), _d(residual), _d(cmpl), _d(cond), _d(dslack), _d(ddual), _d(Qaa), _d(la),
                                      _d(da if da is not None else np.zeros(nv)), _d(steps))
        return rc, cond, dslack, ddual, steps

    g = np.array([w.sign * a[w.index] - bounds[r] for r, w in enumerate(rows)])
    slack, dual = np.zeros(nrow), np.zeros(nrow)
    rc = call(1, slack, dual, np.zeros(nrow), np.zeros(nrow), np.zeros(nv), np.zeros(nv))[0]
    if impact:
        assert rc == 1   # acceleration-level rows do not exist on impact grids (constraints_data.cpp:20-45)
        assert not cr.joint_limit_active(rows, time_stage, impact).any()
        return
    assert rc == 0
    s0, d0 = cr.init_slack_dual(g, barrier)
    assert np.allclose(slack, s0, rtol=1e-14, atol=0) and np.allclose(dual, d0, rtol=1e-14, atol=0)
    # linearize + condense + expand at random positive slack / dual
    slack, dual = rng.uniform(0.1, 2.0, nrow), rng.uniform(0.1, 2.0, nrow)
    residual, cmpl = np.zeros(nrow), np.zeros(nrow)
    Qaa0, la0, da = rng.uniform(0.5, 2.0, nv), rng.uniform(-1, 1, nv), rng.uniform(-1, 1, nv)
    Qaa, la = Qaa0.copy(), la0.copy()
    rc, cond, dslack, ddual, steps = call(2 | 4 | 8, slack.copy(), dual.copy(), residual, cmpl, Qaa, la, da)
    assert rc == 0
    assert np.abs(residual - (g + slack)).max() < 1e-13 and np.abs(cmpl - (slack * dual - barrier)).max() < 1e-13
    la_lin = la0.copy()
    for r, w in enumerate(rows):
        la_lin[w.index] += w.sign * dual[r]
    # the C oracle on records carrying the linearised data
    g0 = Grid(GRID_INTERMEDIATE, 0, 0, 0, 12, 0, 10, time_stage, 0.02)
    gt = Grid(3, 0, 0, 0, 0, 0, 0, 11, 0.0)
    Lo = oracle.layout(dims)
    K, D, N, R = Records(Lo, "kkt"), Records(Lo, "cdd"), Records(Lo, "con"), Records(Lo, "dir")
    kkt, cdd, con, dirs = K.zeros(1, 2), D.zeros(1, 2), N.zeros(1, 2), R.zeros(1, 2)
    D.f(cdd[0, 0], "Qaa")[:] = Qaa0
    D.f(cdd[0, 0], "la")[:] = la_lin
    for name, arr in (("slack", slack), ("dual", dual), ("residual", residual), ("cmpl", cmpl)):
        N.f(con[0, 0], name)[:nrow] = arr
    oracle.pdipm_condense_batch(Lo, [g0, gt], rows, kkt, con, cdd)
    w = dict(Qaa=np.abs(D.f(cdd[0, 0], "Qaa") - Qaa).max(), la=np.abs(D.f(cdd[0, 0], "la") - la).max(),
             cond=np.abs(N.f(con[0, 0], "cond")[:nrow] - cond).max())
    R.f(dirs[0, 0], "daf")[:nv] = da
    st = oracle.pdipm_expand_batch(Lo, [g0, gt], rows, con, dirs, tau)
    w["dslack"] = np.abs(N.f(con[0, 0], "dslack")[:nrow] - dslack).max()
    w["ddual"] = np.abs(N.f(con[0, 0], "ddual")[:nrow] - ddual).max()
    w["steps"] = float(np.abs(st[0] - steps).max())
    print("acceleration limits, C oracle vs the reference sources:", {k: "%.1e" % e for k, e in w.items()}, steps)
    assert max(w.values()) < 1e-12
    assert np.abs(Qaa - Qaa0).max() > 0.05 and steps.min() < 1.0
