"""The inequality rows of the contact path evaluated on the device (rtoc_contact_init_constraints, the linearizeConstraints
part of rtoc_contact_eval_kkt): joint-limit rows (src/constraints/joint_position_lower_limit.cpp:40-77 and its siblings) and
friction cones (src/constraints/friction_cone.cpp:100-191) against a numpy restatement on top of the CPU rigid-body oracle
(frame placements; the Jacobian dg/dq by central differences of g over the tangent space).  Then the ANYmal trot with the
Constraints object of examples/anymal/trot.cpp:131-146 closed on the device.  Parity-unpinned (Pinocchio absent)."""
import numpy as np
import pytest

from robotoc_amd import capi, robot_model as rm
from robotoc_amd.grid import ContactSequence, Event, discretize
from robotoc_amd.types import (BUF_CDD, BUF_CON, BUF_CONE, BUF_KKT, BUF_SOL, GRID_IMPACT, Records, anymal_dims, cone_dgdf_off,
                               joint_limit_rows)
from constraint_restatement import friction_cone_rows
from test_switching_constraint_lin import fd_cols, trot_masks

Q_STAND = np.array([0, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0])
BARRIER = 1.0e-3


def cone_world(mu):
    m = mu / np.sqrt(2.0)
    return np.array([[0, 0, -1], [1, 0, -m], [-1, 0, -m], [0, 1, -m], [0, -1, -m]], dtype=float)


def limits(nu, qmax=2.0, vmax=7.5, umax=40.0, amax=None):
    """bounds of joint_limit_rows: g = sign z - bound <= 0, symmetric limits (amax: with the acceleration rows)"""
    return np.concatenate([np.full(2 * nu, qmax), np.full(2 * nu, vmax), np.full(2 * nu, umax)] + ([np.full(2 * nu, amax)] if amax else []))


@pytest.mark.gpu
@pytest.mark.parametrize("exact,accel", [(False, False), (True, False), (False, True)])
def test_cone_and_joint_limit_rows_against_the_numpy_restatement(oracle, exact, accel):
    """accel: with JointAcceleration{Lower,Upper}Limit rows (RTOC_VAR_A: g = sign a - bound, la += sign dual,
    joint_acceleration_lower_limit.cpp:43-66).  exact: RTOC_OPT_CONE_JACOBIAN -- False: dg/dq as the reference composes it (LOCAL-frame angular Jacobian x world-frame
    force; the restatement tests/constraint_restatement.py is pinned to the reference's sources by
    tests/test_constraints_vs_reference.py), True: the derivative of R_wf(q) f, checked against central differences of g."""
    m = rm.load_named("anymal")
    dims = anymal_dims(nc_max=120 if accel else 96)
    cs = ContactSequence([12, 6, 12], [Event("lift", 0.105), Event("impact", 0.265, impact_dimf=6)])
    grids = discretize(20, 0.4, 0.0, cs)
    n, nv, nq, nu, batch = len(grids), m.nv, m.nq, 12, 2
    masks = trot_masks(grids, [0b1111, 0b1001, 0b1111], [0b0110])
    rng = np.random.default_rng(11)
    feet = np.array([oracle.rbd_contact_position(m, Q_STAND, c) for c in range(4)])
    rot = np.zeros((n, 4, 3, 3))
    for i in range(n):
        for c in range(4):
            rot[i, c] = oracle.rbd_exp6(np.concatenate([np.zeros(3), 0.2 * rng.uniform(-1, 1, 3)]))[0]
    mu = np.array([0.7, 0.6, 0.8, 0.5])
    rows = joint_limit_rows(dims, acceleration=accel)
    bounds = limits(nu, amax=0.8 if accel else None)   # a ~ U(-1, 1): some beyond the limit (slack clipped)
    ctx = capi.Context(dims, n, batch, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_contact_schedule(masks, np.tile(feet[None], (n, 1, 1)), rot.reshape(n, 4, 9))
    ctx.set_constraint_rows(rows)
    ctx.set_friction_cones(4, 3)
    ctx.set_constraint_bounds(bounds, BARRIER, 0.995)
    ctx.set_friction_coefficients(mu)
    ctx.set_cone_jacobian(exact)
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])
    ctx.set_configuration_cost(Q_STAND, np.zeros(nv), np.zeros(12), wq, np.full(nv, 1.0), np.full(nv, 1e-3), np.full(12, 1e-3),
                               10.0 * wq, np.full(nv, 1.0), q_weight_impact=wq, v_weight_impact=np.full(nv, 1.0), dv_weight_impact=np.full(nv, 1e-3))
    ctx.set_initial_state(np.tile(np.concatenate([Q_STAND, np.zeros(nv)]), (batch, 1)))
    S, K, D, N = Records(ctx.L, "sol"), Records(ctx.L, "kkt"), Records(ctx.L, "cdd"), Records(ctx.L, "con")
    sol = S.zeros(batch, n)
    for b in range(batch):
        for i in range(n):
            q = Q_STAND.copy()
            q[:7] = oracle.se3_integrate(Q_STAND[:7], 0.2 * rng.uniform(-1, 1, 6))
            q[7:] += 0.3 * rng.uniform(-1, 1, 12)
            S.f(sol[b, i], "q")[:nq] = q
            S.f(sol[b, i], "v")[:] = 3.0 * rng.uniform(-1, 1, nv)
            S.f(sol[b, i], "a")[:] = rng.uniform(-1, 1, nv)
            S.f(sol[b, i], "u")[:] = 60.0 * rng.uniform(-1, 1, nu)   # some beyond the torque limit: slack clipped
            f = rng.uniform(-1, 1, 12) * 20.0
            f[2::3] = rng.uniform(20, 80, 4)
            S.f(sol[b, i], "f")[:] = f
    ctx.upload(BUF_SOL, sol)
    ctx.contact_init_constraints()
    con0 = ctx.download_records(BUF_CON, "con")
    ctx.contact_eval_kkt()
    con1, kkt1, cdd1 = ctx.download_records(BUF_CON, "con"), ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd")
    stride = ctx.buffer_count(BUF_CONE) // (batch * n)
    cone = ctx.download(BUF_CONE, (batch, n, stride))
    # the same evaluation with the duals removed: the difference is what the rows add to the residuals
    conz = con0.copy()
    N.f(conz, "dual")[:] = 0.0
    ctx.upload(BUF_CON, conz)
    ctx.contact_eval_kkt()
    kkt2, cdd2 = ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd")
    row0 = dims.nc_max - 20
    sb = np.sqrt(BARRIER)
    worst = dict(slack=0.0, dual=0.0, residual=0.0, cmpl=0.0, dgdf=0.0, dgdq=0.0, lq=0.0, lf=0.0, lv=0.0, lu=0.0, la=0.0)
    accel_rows = 0
    for b in range(batch):
        for i in range(n - 1):
            s, g = sol[b, i], grids[i]
            q, v, u, f = S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "u"), S.f(s, "f")
            slack, dual = N.f(con0[b, i], "slack"), N.f(con0[b, i], "dual")
            lx_add, lu_add, lf_add, la_add = np.zeros(2 * nv), np.zeros(nu), np.zeros(12), np.zeros(nv)
            # ---- joint limits ----
            for r, w in enumerate(rows):
                if g.type == GRID_IMPACT or g.time_stage < w.level:
                    assert slack[r] == 0.0 and dual[r] == 0.0
                    continue
                z = q[w.index + 1] if w.var == 0 else (v[w.index] if w.var == 1 else (S.f(s, "a")[w.index] if w.var == 3 else u[w.index]))
                gval = w.sign * z - bounds[r]
                worst["slack"] = max(worst["slack"], abs(slack[r] - max(-gval, sb)))
                worst["dual"] = max(worst["dual"], abs(dual[r] - BARRIER / slack[r]))
                worst["residual"] = max(worst["residual"], abs(N.f(con1[b, i], "residual")[r] - (gval + slack[r])))
                worst["cmpl"] = max(worst["cmpl"], abs(N.f(con1[b, i], "cmpl")[r] - (slack[r] * dual[r] - BARRIER)))
                if w.var == 2:
                    lu_add[w.index] += w.sign * dual[r]
                elif w.var == 3:
                    la_add[w.index] += w.sign * dual[r]
                    accel_rows += 1
                else:
                    lx_add[w.index + (nv if w.var == 1 else 0)] += w.sign * dual[r]
            # ---- friction cones: none on the impact grid (RTOC_OPT_IMPACT_CONES is on by default -> rows there too) ----
            act = [c for c in range(4) if (int(masks[i]) >> c) & 1]
            for k, c in enumerate(act):
                Cl = cone_world(mu[c]) @ rot[i, c].T

                def gfun(qq):
                    return Cl @ (oracle.rbd_contact_placement(m, qq, c)[0] @ f[3 * k:3 * k + 3])

                gval = gfun(q)
                rr = slice(row0 + 5 * k, row0 + 5 * k + 5)
                assert (slack[rr] > 0).all(), (b, i, k, c, g.type, slack[rr])
                worst["slack"] = max(worst["slack"], np.abs(slack[rr] - np.maximum(-gval, sb)).max())
                worst["dual"] = max(worst["dual"], np.abs(dual[rr] - BARRIER / slack[rr]).max())
                worst["residual"] = max(worst["residual"], np.abs(N.f(con1[b, i], "residual")[rr] - (gval + slack[rr])).max())
                worst["cmpl"] = max(worst["cmpl"], np.abs(N.f(con1[b, i], "cmpl")[rr] - (slack[rr] * dual[rr] - BARRIER)).max())
                Rwf = oracle.rbd_contact_placement(m, q, c)[0]
                dgdf = Cl @ Rwf
                if exact:
                    dgdq = fd_cols(lambda e: gfun(oracle.rbd_integrate(m, q, e)), nv)
                else:
                    # world-aligned angular Jacobian of the frame from central differences of its rotation: [w]x = dR R^T
                    dR = fd_cols(lambda e: oracle.rbd_contact_placement(m, oracle.rbd_integrate(m, q, e), c)[0].reshape(-1), nv)
                    ww = np.zeros((3, nv))
                    for jj in range(nv):
                        W = dR[:, jj].reshape(3, 3) @ Rwf.T
                        ww[:, jj] = [W[2, 1], W[0, 2], W[1, 0]]
                    g_r, dgdq, dgdf_r = friction_cone_rows(mu[c], rot[i, c], Rwf, ww, f[3 * k:3 * k + 3], exact_jacobian=False)
                    assert np.abs(g_r - gval).max() < 1e-12 and np.abs(dgdf_r - dgdf).max() < 1e-12
                dev_dgdq = cone[b, i, k * 5 * nv:(k + 1) * 5 * nv].reshape(nv, 5).T
                o = cone_dgdf_off(nv, 4) + 15 * k
                dev_dgdf = cone[b, i, o:o + 15].reshape(3, 5).T
                worst["dgdf"] = max(worst["dgdf"], np.abs(dev_dgdf - dgdf).max())
                worst["dgdq"] = max(worst["dgdq"], np.abs(dev_dgdq - dgdq).max())
                lx_add[:nv] += dev_dgdq.T @ dual[rr]
                lf_add[3 * k:3 * k + 3] += dev_dgdf.T @ dual[rr]
            dlx = K.f(kkt1[b, i], "lx") - K.f(kkt2[b, i], "lx")
            worst["lq"] = max(worst["lq"], np.abs(dlx[:nv] - lx_add[:nv]).max())
            worst["lv"] = max(worst["lv"], np.abs(dlx[nv:] - lx_add[nv:]).max())
            worst["lu"] = max(worst["lu"], np.abs(K.f(kkt1[b, i], "lu") - K.f(kkt2[b, i], "lu") - lu_add).max())
            worst["lf"] = max(worst["lf"], np.abs(D.f(cdd1[b, i], "lf") - D.f(cdd2[b, i], "lf") - lf_add).max())
            worst["la"] = max(worst["la"], np.abs(D.f(cdd1[b, i], "la") - D.f(cdd2[b, i], "la") - la_add).max())
    print("constraint rows, worst deviations:", {k: "%.1e" % e for k, e in worst.items()})
    assert max(worst[k] for k in ("slack", "dual", "residual", "cmpl", "dgdf")) < 1e-11
    assert worst["dgdq"] < 1e-6                                    # central differences
    assert max(worst[k] for k in ("lq", "lv", "lu", "lf", "la")) < 1e-9  # differences of O(100) residual entries
    assert accel_rows == (2 * nu * batch * sum(1 for g in grids[:-1] if g.type != GRID_IMPACT) if accel else 0)
    ctx.close()


@pytest.mark.gpu
def test_anymal_trot_with_joint_limits_and_friction_cones_converges_on_the_device(oracle):
    """examples/anymal/trot.cpp's Constraints object (six joint-limit components + FrictionCone, barrier 1e-3, fraction to
    boundary 0.995; no ImpactFrictionCone -> RTOC_OPT_IMPACT_CONES off) on the trot's 47 grid points, every part of the
    iteration on the device.  Limits chosen so that rows are active at the solution: a torque limit below what the stance
    knees would like, a small friction coefficient while the feet step forward."""
    from robotoc_amd.problems import config_anymal_trot
    m = rm.load_named("anymal")
    dims, grids, _ = config_anymal_trot()
    n, nv, nq, nu, batch = len(grids), m.nv, m.nq, 12, 2
    masks = trot_masks(grids, [0b1111, 0b1001, 0b1111, 0b0110, 0b1111], [0b0110, 0b1001])
    feet = np.array([oracle.rbd_contact_position(m, Q_STAND, c) for c in range(4)])
    pos = np.tile(feet[None], (n, 1, 1))
    impacts = [i for i, g in enumerate(grids) if g.type == GRID_IMPACT]
    pos[impacts[0]:, [1, 2], 0] += 0.05
    pos[impacts[1]:, [0, 3], 0] += 0.05
    rows = joint_limit_rows(dims)
    mu = np.full(4, 0.2)
    ctx = capi.Context(dims, n, batch, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_contact_schedule(masks, pos)
    ctx.set_constraint_rows(rows)
    ctx.set_friction_cones(4, 3)
    ctx.set_impact_cones(False)
    mass = sum(m.mass[i] for i in range(m.njoints))
    # gravity compensation on two feet needs more knee torque than on four: put the limit in between
    q0 = Q_STAND
    f4 = np.concatenate([oracle.rbd_contact_placement(m, q0, c)[0].T @ np.array([0.0, 0.0, 9.81 * mass / 4]) for c in range(4)])
    u4 = oracle.rbd_eval(m, 0, q0, np.zeros(nv), np.zeros(nv), f4, np.zeros(12), 0b1111, feet.reshape(-1))[6:nv]
    umax = 1.25 * np.abs(u4).max()
    bounds = limits(nu, qmax=1.2, vmax=3.0, umax=umax)
    ctx.set_constraint_bounds(bounds, BARRIER, 0.995)
    ctx.set_friction_coefficients(mu)
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])
    q_ref = Q_STAND.copy()
    q_ref[0] += 0.15   # the base is asked forward: tangential contact forces, against a small friction coefficient
    ctx.set_configuration_cost(q_ref, np.zeros(nv), np.zeros(12), wq, np.full(nv, 1.0), np.full(nv, 1e-3), np.full(12, 1e-3),
                               10.0 * wq, np.full(nv, 1.0), q_weight_impact=wq, v_weight_impact=np.full(nv, 1.0), dv_weight_impact=np.full(nv, 1e-3))
    x0 = np.tile(np.concatenate([Q_STAND, np.zeros(nv)]), (batch, 1))
    x0[1, :7] = oracle.se3_integrate(Q_STAND[:7], np.array([0.01, 0.005, -0.005, 0.0, 0.02, 0.01]))
    ctx.set_initial_state(x0)
    S, N = Records(ctx.L, "sol"), Records(ctx.L, "con")
    sol = S.zeros(batch, n)
    for b in range(batch):
        qb = x0[b, :nq]
        for i in range(n):
            act = [c for c in range(4) if (int(masks[i]) >> c) & 1]
            S.f(sol[b, i], "q")[:nq] = qb
            if act and grids[i].type != GRID_IMPACT:
                S.f(sol[b, i], "f")[:3 * len(act)] = np.concatenate([oracle.rbd_contact_placement(m, qb, c)[0].T @ np.array([0.0, 0.0, 9.81 * mass / len(act)]) for c in act])
    ctx.upload(BUF_SOL, sol)
    ctx.contact_init_constraints()
    hist = []
    for it in range(150):
        hist.append(ctx.contact_update_solution(0.995))
        if hist[-1].max() < 1e-8:
            break
    hist = np.array(hist)
    print("KKT error per iteration (worst instance):", ["%.1e" % e for e in hist.max(axis=1)])
    print("umax %.2f (four-feet hold %.2f)" % (umax, np.abs(u4).max()))
    assert (ctx.status() == 0).all() and hist[-1].max() < 1e-7
    sol, con = ctx.download_records(BUF_SOL, "sol"), ctx.download_records(BUF_CON, "con")
    row0 = dims.nc_max - 20
    worst = dict(g_max=-np.inf, central_path=0.0, slack_gap=0.0, IDC=0.0)
    near = dict(torque=0, cone=0)
    for b in range(batch):
        for i in range(n - 1):
            s, g = sol[b, i], grids[i]
            q, v, a, u, f = S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "a"), S.f(s, "u"), S.f(s, "f")
            slack, dual = N.f(con[b, i], "slack"), N.f(con[b, i], "dual")
            r = oracle.rbd_eval(m, int(g.type == GRID_IMPACT), q, v, a, f[:12], u[:12], int(masks[i]), pos[i].reshape(-1))
            worst["IDC"] = max(worst["IDC"], np.abs(r).max())
            for k, w in enumerate(rows):
                if g.time_stage < w.level:
                    continue
                z = q[w.index + 1] if w.var == 0 else (v[w.index] if w.var == 1 else u[w.index])
                gval = w.sign * z - bounds[k]
                worst["g_max"] = max(worst["g_max"], gval)
                worst["slack_gap"] = max(worst["slack_gap"], abs(gval + slack[k]))
                worst["central_path"] = max(worst["central_path"], abs(slack[k] * dual[k] - BARRIER))
                near["torque"] += int(w.var == 2 and gval > -0.05 * umax)
            if g.type == GRID_IMPACT:
                continue
            act = [c for c in range(4) if (int(masks[i]) >> c) & 1]
            for k, c in enumerate(act):
                gval = cone_world(mu[c]) @ (oracle.rbd_contact_placement(m, q, c)[0] @ f[3 * k:3 * k + 3])
                rr = slice(row0 + 5 * k, row0 + 5 * k + 5)
                worst["g_max"] = max(worst["g_max"], gval.max())
                worst["slack_gap"] = max(worst["slack_gap"], np.abs(gval + slack[rr]).max())
                worst["central_path"] = max(worst["central_path"], np.abs(slack[rr] * dual[rr] - BARRIER).max())
                near["cone"] += int((gval[1:] > -0.05 * mu[c] * f[3 * k + 2]).any())
    print("converged constrained trot:", worst, "rows within 5 % of their bound:", near)
    assert worst["g_max"] < 0.0 and worst["slack_gap"] < 1e-7 and worst["central_path"] < 1e-7 and worst["IDC"] < 1e-7
    assert near["torque"] > 0 or near["cone"] > 0   # the limits shape the solution
    ctx.close()


Q_ICUB = np.array([0, 0, 0.592, 0, 0, 1, 0,
                   0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726, 0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726,
                   0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0])   # examples/icub/python/jump_sto.py:21-26


@pytest.mark.gpu
@pytest.mark.parametrize("cones", ["friction", "wrench"])
def test_icub_on_two_soles_with_limits_and_cones_converges_on_the_device(oracle, cones):
    """iCub (nv = 35, two surface contacts: six contact rows each, Log6 placement error) standing on both soles, N = 15:
    ConfigurationSpaceCost, the six joint-limit components and either FrictionCone on the first three wrench components
    (examples/icub/python/jump_sto.py:60-68) or ContactWrenchCone (17 rows per sole: friction pyramid, centre of pressure
    inside the 0.2 x 0.1 sole, yaw torque) -- every part of the iteration on the device."""
    from robotoc_amd.grid import uniform_grid
    from robotoc_amd.types import icub_dims
    m = rm.load_named("icub")
    nv, nq, nu = m.nv, m.nq, m.nv - 6
    cone_rows = 10 if cones == "friction" else 34
    dims = icub_dims(nv, nc_max=(6 * nu + cone_rows + 7) & ~7)
    N, dt, batch = 15, 0.02, 2
    grids = uniform_grid(N, dt, dimf=12)
    n = len(grids)
    placements = [oracle.rbd_contact_placement(m, Q_ICUB, c) for c in range(2)]
    pos = np.tile(np.array([p for _, p in placements])[None], (n, 1, 1))
    rot = np.tile(np.array([R.reshape(9) for R, _ in placements])[None], (n, 1, 1))
    ctx = capi.Context(dims, n, batch, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_contact_schedule(np.full(n, 0b11, dtype=np.uint32), pos, rot)
    rows = joint_limit_rows(dims)
    ctx.set_constraint_rows(rows)
    mu = 0.6
    if cones == "friction":
        ctx.set_friction_cones(2, 6)
    else:
        ctx.set_wrench_cones(2)
    bounds = limits(nu, qmax=2.5, vmax=5.0, umax=60.0)
    ctx.set_constraint_bounds(bounds, BARRIER, 0.995)
    if cones == "friction":
        ctx.set_friction_coefficients(np.full(2, mu))
    else:
        ctx.set_wrench_cone_params(np.array([[0.1, 0.05, mu], [0.1, 0.05, mu]]))
    q_ref = Q_ICUB.copy()
    q_ref[2] -= 0.03   # squat a little
    wq = np.concatenate([np.full(6, 10.0), np.full(nu, 0.1)])
    ctx.set_configuration_cost(q_ref, np.zeros(nv), np.zeros(nu), wq, np.full(nv, 0.1), np.full(nv, 1e-3), np.full(nu, 1e-4), 10.0 * wq, np.full(nv, 0.1))
    x0 = np.tile(np.concatenate([Q_ICUB, np.zeros(nv)]), (batch, 1))
    x0[1, nq:] = 0.02 * np.random.default_rng(8).uniform(-1, 1, nv)
    ctx.set_initial_state(x0)
    S, Nn = Records(ctx.L, "sol"), Records(ctx.L, "con")
    sol = S.zeros(batch, n)
    mass = sum(m.mass[i] for i in range(m.njoints))
    f0 = np.concatenate([np.concatenate([R.T @ np.array([0.0, 0.0, 9.81 * mass / 2]), np.zeros(3)]) for R, _ in placements])
    for b in range(batch):
        S.f(sol[b], "q")[:, :nq] = x0[b, :nq]
        S.f(sol[b], "f")[:, :12] = f0
    ctx.upload(BUF_SOL, sol)
    ctx.contact_init_constraints()
    hist = []
    for it in range(120):
        hist.append(ctx.contact_update_solution(0.995))
        if hist[-1].max() < 1e-8:
            break
    hist = np.array(hist)
    print("iCub, %s cones: KKT error per iteration (worst instance):" % cones, ["%.1e" % e for e in hist.max(axis=1)])
    assert (ctx.status() == 0).all() and hist[-1].max() < 1e-6
    sol, con = ctx.download_records(BUF_SOL, "sol"), ctx.download_records(BUF_CON, "con")
    row0 = dims.nc_max - cone_rows
    worst = dict(g_max=-np.inf, slack_gap=0.0, central_path=0.0, IDC=0.0)
    A = None
    if cones == "wrench":
        A = np.zeros((17, 6), order="F")
        capi.lib().rtoc_wrench_cone_matrix(0.1, 0.05, mu, A.ctypes.data_as(capi.C.POINTER(capi.C.c_double)))
    for b in range(batch):
        for i in range(n - 1):
            s = sol[b, i]
            q, v, a, u, f = S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "a"), S.f(s, "u"), S.f(s, "f")
            slack, dual = Nn.f(con[b, i], "slack"), Nn.f(con[b, i], "dual")
            r = oracle.rbd_eval(m, 0, q, v, a, f[:12], u[:nu], 0b11, pos[i].reshape(-1), rot[i].reshape(-1))
            worst["IDC"] = max(worst["IDC"], np.abs(r).max())
            for k in range(2):
                if cones == "friction":
                    # the cone stands on the contact surface: ContactStatus::contactRotation = the desired sole rotation
                    gval = cone_world(mu) @ rot[i, k].reshape(3, 3).T @ (oracle.rbd_contact_placement(m, q, k)[0] @ f[6 * k:6 * k + 3])
                    rr = slice(row0 + 5 * k, row0 + 5 * k + 5)
                else:
                    gval = A @ f[6 * k:6 * k + 6]
                    rr = slice(row0 + 17 * k, row0 + 17 * k + 17)
                worst["g_max"] = max(worst["g_max"], gval.max())
                worst["slack_gap"] = max(worst["slack_gap"], np.abs(gval + slack[rr]).max())
                worst["central_path"] = max(worst["central_path"], np.abs(slack[rr] * dual[rr] - BARRIER).max())
            for k, w in enumerate(rows):
                if grids[i].time_stage < w.level:
                    continue
                z = q[w.index + 1] if w.var == 0 else (v[w.index] if w.var == 1 else u[w.index])
                worst["g_max"] = max(worst["g_max"], w.sign * z - bounds[k])
                gap = abs(w.sign * z - bounds[k] + slack[k])
                assert gap < 1e-6, ("joint-limit row", k, w.var, w.index, w.sign, w.level, i, z, bounds[k], slack[k])
                worst["slack_gap"] = max(worst["slack_gap"], gap)
    print("converged iCub stand:", worst)
    assert worst["g_max"] < 0.0 and worst["slack_gap"] < 1e-6 and worst["central_path"] < 1e-6 and worst["IDC"] < 1e-6
    ctx.close()
