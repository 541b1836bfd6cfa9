"""OCPSolver::updateSolution (src/solver/ocp_solver.cpp:111-145) closed on the device for a floating-base robot in contact:
ANYmal on its four feet, ConfigurationSpaceCost, no inequality rows, no discrete events.  Cost (SE(3) difference of the base
included), state equation, contact-dynamics linearisation, condensation, Riccati sweep, expansion and the solution update
on the manifold all run in rtoc_contact_update_solution; the host only reads the KKT error.  Checks: Newton convergence from
the reference examples' kind of initial guess (the initial state on every grid point, gravity-compensating contact forces);
the converged trajectory against the CPU restatement -- inverse dynamics with the contact forces, Baumgarte contact rows,
state equation on SE(3), initial state.  The rigid-body parts are parity-unpinned (Pinocchio absent)."""
import numpy as np
import pytest

from robotoc_amd import capi, robot_model as rm
from robotoc_amd.grid import uniform_grid
from robotoc_amd.types import BUF_SOL, Records, anymal_dims

Q_STAND = np.array([0, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0])  # examples/anymal


@pytest.mark.gpu
def test_anymal_standing_solver_iterations_converge_on_the_device(oracle):
    m = rm.load_named("anymal")
    dims = anymal_dims()
    N, dt, batch = 20, 0.02, 4
    grids = uniform_grid(N, dt, dimf=12)
    n, nv, nq = len(grids), m.nv, m.nq
    ctx = capi.Context(dims, n, batch, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    feet = np.array([oracle.rbd_contact_position(m, Q_STAND, c) for c in range(4)])
    ctx.set_contact_schedule(np.full(n, 0b1111, dtype=np.uint32), np.tile(feet[None], (n, 1, 1)))
    # track a base placement 3 cm forward, 2 cm down and slightly pitched, joints near the standing pose
    q_ref = Q_STAND.copy()
    q_ref[:7] = oracle.se3_integrate(Q_STAND[:7], np.array([0.03, 0.0, -0.02, 0.0, 0.05, 0.0]))
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 0.1)])
    ctx.set_configuration_cost(q_ref, np.zeros(nv), np.zeros(12), wq, np.full(nv, 1.0), np.full(nv, 1e-3), np.full(12, 1e-3),
                               10.0 * wq, np.full(nv, 1.0))
    rng = np.random.default_rng(31)
    x0 = np.tile(np.concatenate([Q_STAND, np.zeros(nv)]), (batch, 1))
    for b in range(batch):  # a different initial state per instance: the base nudged, joints moved so that the feet stay put is NOT required
        x0[b, :7] = oracle.se3_integrate(Q_STAND[:7], 0.01 * rng.uniform(-1, 1, 6))
        x0[b, nq:] = 0.02 * rng.uniform(-1, 1, nv)
    ctx.set_initial_state(x0)
    S = Records(ctx.L, "sol")
    sol = S.zeros(batch, n)
    mass = sum(m.mass[i] for i in range(m.njoints))
    for b in range(batch):
        q0 = x0[b, :nq]
        f = np.concatenate([oracle.rbd_contact_placement(m, q0, c)[0].T @ np.array([0.0, 0.0, 9.81 * mass / 4]) for c in range(4)])
        u = oracle.rbd_eval(m, 0, q0, np.zeros(nv), np.zeros(nv), f, np.zeros(12), 0b1111, feet.reshape(-1))[6:nv]
        for i in range(n):
            S.f(sol[b, i], "q")[:nq] = q0
            S.f(sol[b, i], "f")[:12] = f
            S.f(sol[b, i], "u")[:12] = u
    ctx.upload(BUF_SOL, sol)
    hist = []
    for it in range(60):
        hist.append(ctx.contact_update_solution())
        if hist[-1].max() < 1e-9:
            break
    hist = np.array(hist)
    print("KKT error per iteration (worst instance):", ["%.1e" % e for e in hist.max(axis=1)])
    assert (ctx.status() == 0).all()
    assert hist[-1].max() < 1e-7 and hist[-1].max() < 1e-8 * hist[0].max()
    # ---- the converged trajectory, by the CPU restatement ----
    sol = ctx.download_records(BUF_SOL, "sol")
    worst = dict(IDC=0.0, Fx=0.0, x0=0.0)
    for b in range(batch):
        q, v = S.f(sol[b, 0], "q")[:nq], S.f(sol[b, 0], "v")
        worst["x0"] = max(worst["x0"], np.abs(oracle.se3_difference(q[:7], x0[b, :7])).max(), np.abs(q[7:] - x0[b, 7:nq]).max(), np.abs(v - x0[b, nq:]).max())
        for i in range(n - 1):
            s, sn = sol[b, i], sol[b, i + 1]
            q, v, a, u, f = S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "a"), S.f(s, "u")[:12], S.f(s, "f")[:12]
            qn, vn = S.f(sn, "q")[:nq], S.f(sn, "v")
            worst["IDC"] = max(worst["IDC"], np.abs(oracle.rbd_eval(m, 0, q, v, a, f, u, 0b1111, feet.reshape(-1))).max())
            Fq = np.concatenate([oracle.se3_difference(qn[:7], q[:7]), q[7:] - qn[7:]]) + dt * v
            worst["Fx"] = max(worst["Fx"], np.abs(Fq).max(), np.abs(v + dt * a - vn).max())
    print("converged trajectory, worst residuals by the CPU restatement:", worst)
    assert worst["IDC"] < 1e-7 and worst["Fx"] < 1e-8 and worst["x0"] < 1e-8
    # the base moves towards the reference placement, the feet stay where they were
    qT = S.f(sol[0, n - 1], "q")[:nq]
    assert np.linalg.norm(oracle.se3_difference(q_ref[:7], qT[:7])) < 0.8 * np.linalg.norm(oracle.se3_difference(q_ref[:7], x0[0, :7]))
    assert max(np.abs(oracle.rbd_contact_position(m, qT, c) - feet[c]).max() for c in range(4)) < 5e-3
    ctx.close()


@pytest.mark.gpu
def test_anymal_lifting_two_feet_converges_on_the_device(oracle):
    """The same OCP with a lift event half way (ContactSequence: four feet -> the LF / RH diagonal): lift grid, a change of
    contact dimension along the horizon, swing legs free.  (Touch-downs need the switching constraint, which is not part of
    the device-side linearisation.)"""
    from robotoc_amd.grid import ContactSequence, Event, discretize
    from robotoc_amd.types import GRID_LIFT
    m = rm.load_named("anymal")
    dims = anymal_dims()
    N, dt, batch = 20, 0.02, 2
    grids = discretize(N, N * dt, 0.0, ContactSequence([12, 6], [Event("lift", 0.205)]))
    n, nv, nq = len(grids), m.nv, m.nq
    assert any(g.type == GRID_LIFT for g in grids) and {g.dimf for g in grids[:-1]} == {12, 6}
    ctx = capi.Context(dims, n, batch, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    feet = np.array([oracle.rbd_contact_position(m, Q_STAND, c) for c in range(4)])
    masks = np.array([0b1111 if g.dimf == 12 else (0b1001 if g.dimf == 6 else 0) for g in grids], dtype=np.uint32)
    ctx.set_contact_schedule(masks, np.tile(feet[None], (n, 1, 1)))
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])
    ctx.set_configuration_cost(Q_STAND, np.zeros(nv), np.zeros(12), wq, np.full(nv, 1.0), np.full(nv, 1e-3), np.full(12, 1e-3),
                               10.0 * wq, np.full(nv, 1.0))
    x0 = np.tile(np.concatenate([Q_STAND, np.zeros(nv)]), (batch, 1))
    x0[1, :7] = oracle.se3_integrate(Q_STAND[:7], np.array([0.01, 0.005, -0.005, 0.0, 0.02, 0.01]))
    ctx.set_initial_state(x0)
    S = Records(ctx.L, "sol")
    sol = S.zeros(batch, n)
    mass = sum(m.mass[i] for i in range(m.njoints))
    for b in range(batch):
        q0 = x0[b, :nq]
        for i in range(n):
            act = [c for c in range(4) if (int(masks[i]) >> c) & 1]
            S.f(sol[b, i], "q")[:nq] = q0
            if act:
                S.f(sol[b, i], "f")[:3 * len(act)] = np.concatenate([oracle.rbd_contact_placement(m, q0, c)[0].T @ np.array([0.0, 0.0, 9.81 * mass / len(act)]) for c in act])
    ctx.upload(BUF_SOL, sol)
    hist = []
    for it in range(80):
        hist.append(ctx.contact_update_solution())
        if hist[-1].max() < 1e-9:
            break
    hist = np.array(hist)
    print("KKT error per iteration (worst instance):", ["%.1e" % e for e in hist.max(axis=1)])
    assert (ctx.status() == 0).all() and hist[-1].max() < 1e-7
    sol = ctx.download_records(BUF_SOL, "sol")
    worst = 0.0
    for b in range(batch):
        for i in range(n - 1):
            s = sol[b, i]
            act = int(masks[i])
            nf = 3 * bin(act).count("1")
            r = oracle.rbd_eval(m, 0, S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "a"), S.f(s, "f")[:12], S.f(s, "u")[:12], act, feet.reshape(-1))
            assert r.size == nv + nf
            worst = max(worst, np.abs(r).max())
    print("dynamics + contact rows of the converged trajectory:", worst)
    assert worst < 1e-7
    ctx.close()


@pytest.mark.gpu
def test_anymal_trot_with_touch_downs_converges_on_the_device(oracle):
    """BASELINE configs[1]'s contact sequence (examples/anymal/trot.cpp:162-190: stand, LH / RF swing, stand, LF / RH swing,
    stand; N = 40, 47 grid points) closed on the device: lift grids, impact grids with the impact dynamics, and the
    switching constraints two grid points ahead of each touch-down (rtoc_contact_eval_kkt's switching-constraint rows).
    Each swing foot touches down 3 cm ahead of where it lifted off; the cost tracks the standing configuration."""
    from robotoc_amd.problems import config_anymal_trot
    from robotoc_amd.types import GRID_IMPACT
    from test_switching_constraint_lin import trot_masks
    m = rm.load_named("anymal")
    dims, grids, _ = config_anymal_trot()
    n, nv, nq, batch = len(grids), m.nv, m.nq, 2
    assert n == 47 and sum(g.switching_constraint for g in grids) == 2
    # contact order of the model: LF, LH, RF, RH
    masks = trot_masks(grids, [0b1111, 0b1001, 0b1111, 0b0110, 0b1111], [0b0110, 0b1001])
    for g, k in zip(grids, masks):
        assert 3 * bin(int(k)).count("1") == g.dimf
    ctx = capi.Context(dims, n, batch, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    feet = np.array([oracle.rbd_contact_position(m, Q_STAND, c) for c in range(4)])
    # each swing foot touches down 3 cm ahead of where it lifted off
    pos = np.tile(feet[None], (n, 1, 1))
    impacts = [i for i, g in enumerate(grids) if g.type == GRID_IMPACT]
    pos[impacts[0]:, [1, 2], 0] += 0.03
    pos[impacts[1]:, [0, 3], 0] += 0.03
    ctx.set_contact_schedule(masks, pos)
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])
    ctx.set_configuration_cost(Q_STAND, np.zeros(nv), np.zeros(12), wq, np.full(nv, 1.0), np.full(nv, 1e-3), np.full(12, 1e-3),
                               10.0 * wq, np.full(nv, 1.0), q_weight_impact=wq, v_weight_impact=np.full(nv, 1.0), dv_weight_impact=np.full(nv, 1e-3))
    x0 = np.tile(np.concatenate([Q_STAND, np.zeros(nv)]), (batch, 1))
    x0[1, :7] = oracle.se3_integrate(Q_STAND[:7], np.array([0.01, 0.005, -0.005, 0.0, 0.02, 0.01]))
    x0[1, nq:] = 0.02 * np.random.default_rng(3).uniform(-1, 1, nv)
    ctx.set_initial_state(x0)
    S = Records(ctx.L, "sol")
    sol = S.zeros(batch, n)
    mass = sum(m.mass[i] for i in range(m.njoints))
    for b in range(batch):
        q0 = x0[b, :nq]
        for i in range(n):
            act = [c for c in range(4) if (int(masks[i]) >> c) & 1]
            S.f(sol[b, i], "q")[:nq] = q0
            if act and grids[i].type != GRID_IMPACT:
                S.f(sol[b, i], "f")[:3 * len(act)] = np.concatenate([oracle.rbd_contact_placement(m, q0, c)[0].T @ np.array([0.0, 0.0, 9.81 * mass / len(act)]) for c in act])
    ctx.upload(BUF_SOL, sol)
    hist = []
    for it in range(100):
        hist.append(ctx.contact_update_solution())
        if hist[-1].max() < 1e-9:
            break
    hist = np.array(hist)
    print("KKT error per iteration (worst instance):", ["%.1e" % e for e in hist.max(axis=1)])
    assert (ctx.status() == 0).all() and hist[-1].max() < 1e-7
    sol = ctx.download_records(BUF_SOL, "sol")
    worst = dict(IDC=0.0, impact=0.0, switching=0.0, Fx=0.0)
    for b in range(batch):
        for i in range(n - 1):
            s, sn, g = sol[b, i], sol[b, i + 1], grids[i]
            act = int(masks[i])
            q, v, a = S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "a")
            qn, vn = S.f(sn, "q")[:nq], S.f(sn, "v")
            r = oracle.rbd_eval(m, int(g.type == GRID_IMPACT), q, v, a, S.f(s, "f")[:12], S.f(s, "u")[:12], act, pos[i].reshape(-1))
            key = "impact" if g.type == GRID_IMPACT else "IDC"
            worst[key] = max(worst[key], np.abs(r).max())
            if g.type == GRID_IMPACT:   # q+ = q, v+ = v + dv; the touching feet at their positions (impact_dynamics rows carry velocity only)
                worst["Fx"] = max(worst["Fx"], np.abs(oracle.se3_difference(qn[:7], q[:7])).max(), np.abs(q[7:] - qn[7:]).max(), np.abs(v + a - vn).max())
            else:
                Fq = np.concatenate([oracle.se3_difference(qn[:7], q[:7]), q[7:] - qn[7:]]) + g.dt * v
                worst["Fx"] = max(worst["Fx"], np.abs(Fq).max(), np.abs(v + g.dt * a - vn).max())
            if g.switching_constraint:
                dt1, dt2 = g.dt, grids[i + 1].dt
                qp = oracle.rbd_integrate(m, q, (dt1 + dt2) * v + dt1 * dt2 * a)
                imp = [c for c in range(4) if (int(masks[i + 2]) >> c) & 1]
                worst["switching"] = max(worst["switching"], max(np.abs(oracle.rbd_contact_position(m, qp, c) - pos[i + 2, c]).max() for c in imp))
    print("converged trot, worst residuals by the CPU restatement:", worst)
    assert worst["IDC"] < 1e-7 and worst["impact"] < 1e-7 and worst["Fx"] < 1e-8 and worst["switching"] < 1e-8
    # the swing feet are free in between: they leave the ground or stay, but the stance feet do not move
    for i in range(n - 1):
        q = S.f(sol[0, i], "q")[:nq]
        for c in range(4):
            if (int(masks[i]) >> c) & 1 and grids[i].type != GRID_IMPACT:
                assert np.abs(oracle.rbd_contact_position(m, q, c) - pos[i, c]).max() < 5e-3
    ctx.close()


@pytest.mark.gpu
def test_icub_hop_with_surface_contacts_converges_on_the_device(oracle):
    """BASELINE configs[3]'s robot and horizon (iCub, nv = 35, N = 30) through stand - flight - stand: lift grid, a flight
    phase without contact rows, the touch-down of both soles with its switching constraint (Log6 placement rows of surface
    contacts, 12 of them), the impact dynamics -- cost and dynamics only (Gauss-Newton iterations without a line search: the
    inequality rows make this particular problem stall; they are covered on the stand, tests/test_contact_constraints.py)."""
    from robotoc_amd.grid import ContactSequence, Event, contact_masks, discretize
    from robotoc_amd.types import GRID_IMPACT, icub_dims
    from test_switching_constraint_lin import Q_ICUB
    m = rm.load_named("icub")
    nv, nq, nu = m.nv, m.nq, m.nv - 6
    dims = icub_dims(nv)
    grids = discretize(30, 0.6, 0.0, ContactSequence([12, 0, 12], [Event("lift", 0.25), Event("impact", 0.36, impact_dimf=12)]))
    n = len(grids)
    masks = contact_masks(grids, [0b11, 0, 0b11], [0b11])
    place = [oracle.rbd_contact_placement(m, Q_ICUB, c) for c in range(2)]
    pos = np.tile(np.array([p for _, p in place])[None], (n, 1, 1))
    rot = np.tile(np.array([R.reshape(9) for R, _ in place])[None], (n, 1, 1))
    ctx = capi.Context(dims, n, 1, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_contact_schedule(masks, pos, rot)
    wq = np.concatenate([np.full(6, 10.0), np.full(nu, 0.1)])
    ctx.set_configuration_cost(Q_ICUB, np.zeros(nv), np.zeros(nu), wq, np.full(nv, 0.1), np.full(nv, 1e-3), np.full(nu, 1e-4), 10 * wq, np.full(nv, 0.1),
                               q_weight_impact=wq, v_weight_impact=np.full(nv, 0.1), dv_weight_impact=np.full(nv, 1e-3))
    ctx.set_initial_state(np.concatenate([Q_ICUB, np.zeros(nv)])[None])
    S = Records(ctx.L, "sol")
    sol = S.zeros(1, n)
    mass = sum(m.mass[i] for i in range(m.njoints))
    f0 = np.concatenate([np.concatenate([R.T @ np.array([0, 0, 9.81 * mass / 2]), np.zeros(3)]) for R, _ in place])
    S.f(sol[0], "q")[:, :nq] = Q_ICUB
    for i in range(n):
        if masks[i] and grids[i].type != GRID_IMPACT:
            S.f(sol[0, i], "f")[:12] = f0
    ctx.upload(BUF_SOL, sol)
    hist = []
    for it in range(150):
        hist.append(ctx.contact_update_solution()[0])
        if hist[-1] < 1e-8:
            break
    print("iCub hop: %d iterations," % len(hist), ["%.1e" % e for e in hist[:4]], "...", ["%.1e" % e for e in hist[-3:]])
    assert (ctx.status() == 0).all() and hist[-1] < 1e-8
    sol = ctx.download_records(BUF_SOL, "sol")[0]
    worst = dict(IDC=0.0, switching=0.0, air=np.inf)
    for i in range(n - 1):
        s, g = sol[i], grids[i]
        q, v, a = S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "a")
        r = oracle.rbd_eval(m, int(g.type == GRID_IMPACT), q, v, a, S.f(s, "f")[:12], S.f(s, "u")[:nu], int(masks[i]), pos[i].reshape(-1), rot[i].reshape(-1))
        worst["IDC"] = max(worst["IDC"], np.abs(r).max())
        if g.switching_constraint:
            dt1, dt2 = g.dt, grids[i + 1].dt
            qp = oracle.rbd_integrate(m, q, (dt1 + dt2) * v + dt1 * dt2 * a)
            for c in range(2):
                R, p = oracle.rbd_contact_placement(m, qp, c)
                Rd = rot[i + 2, c].reshape(3, 3)
                worst["switching"] = max(worst["switching"], np.abs(oracle.rbd_log6(Rd.T @ R, Rd.T @ (p - pos[i + 2, c]))).max())
    print("converged hop, worst residuals by the CPU restatement:", worst)
    assert worst["IDC"] < 1e-6 and worst["switching"] < 1e-7
    ctx.close()
