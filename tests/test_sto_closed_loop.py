"""BASELINE configs[2] solved end to end on the device: the ANYmal jump with switching-time optimisation of the reference's
examples/anymal/python/jump_sto.py at N = 40 (stand, flight, stand; lift-off and touch-down times optimised; ConfigurationSpaceCost;
six joint-limit components + FrictionCone; minimum dwell times) through robotoc_amd.solver.OCPSolver -- the reference's
OCPSolver::solve loop (src/solver/ocp_solver.cpp:148-225: STO regularisation schedule, updateSolution with sto_.evalKKT /
computeStepSizes / integrateSolution, mesh refinement with solution interpolation) over rtoc_contact_update_solution.
The single iteration is pinned to the reference's sources by tests/test_golden_ref.py (STO fixture); here: convergence of the
whole solve, the event times it finds, and the converged trajectory re-evaluated by the CPU restatement with the time steps
that belong to the optimised event times (rigid-body side parity-unpinned, Pinocchio absent)."""
import numpy as np
import pytest

from robotoc_amd.types import GRID_IMPACT, Records


def _residuals(oracle, solver, b, sol):
    m, grids, masks = solver.model, solver.grids, solver.masks
    S = Records(solver.ctx.L, "sol")
    nq, nv = m.nq, m.nv
    t, dt = solver.grid_times(b)
    n = len(grids)
    pos, phase = np.zeros((n, 4, 3)), 0
    for i, g in enumerate(grids):
        if g.type in (1, 2):
            phase += 1
        pos[i] = solver.plan.phase_positions[min(phase, 2)]
    worst = dict(IDC=0.0, impact=0.0, switching=0.0, Fx=0.0)
    for i in range(n - 1):
        s, sn, g = sol[i], sol[i + 1], grids[i]
        act = int(masks[i])
        q, v, a = S.f(s, "q")[:nq], S.f(s, "v"), S.f(s, "a")
        qn, vn = S.f(sn, "q")[:nq], S.f(sn, "v")
        r = oracle.rbd_eval(m, int(g.type == GRID_IMPACT), q, v, a, S.f(s, "f")[:12], S.f(s, "u")[:12], act, pos[i].reshape(-1))
        key = "impact" if g.type == GRID_IMPACT else "IDC"
        worst[key] = max(worst[key], np.abs(r).max())
        if g.type == GRID_IMPACT:
            worst["Fx"] = max(worst["Fx"], np.abs(oracle.se3_difference(qn[:7], q[:7])).max(), np.abs(q[7:] - qn[7:]).max(), np.abs(v + a - vn).max())
        else:
            Fq = np.concatenate([oracle.se3_difference(qn[:7], q[:7]), q[7:] - qn[7:]]) + dt[i] * v
            worst["Fx"] = max(worst["Fx"], np.abs(Fq).max(), np.abs(v + dt[i] * a - vn).max())
        if g.switching_constraint:
            qp = oracle.rbd_integrate(m, q, (dt[i] + dt[i + 1]) * v + dt[i] * dt[i + 1] * a)
            imp = [c for c in range(4) if (int(masks[i + 2]) >> c) & 1]
            worst["switching"] = max(worst["switching"], max(np.abs(oracle.rbd_contact_position(m, qp, c) - pos[i + 2, c]).max() for c in imp))
    return worst, t, dt


@pytest.mark.gpu
def test_anymal_jump_with_switching_time_optimisation_solves_on_the_device(oracle):
    from robotoc_amd import problems_jump as pj
    solver, x0, info = pj.anymal_jump_sto_solver(batch=1)
    try:
        ts0 = solver.event_times.copy()
        st = solver.solve(0.0, x0)
        hist = np.array([e.max() for e in st.kkt_error])
        print("jump with STO: %d iterations, mesh refinement at %s, KKT %s ... %s" % (st.iter, st.mesh_refinement_iter, ["%.1e" % e for e in hist[:3]],
                                                                                      ["%.1e" % e for e in hist[-3:]]))
        print("event times %s -> %s" % (ts0[0], solver.event_times[0]))
        assert st.convergence and hist[-1] < 1e-7 and (solver.ctx.status() == 0).all()
        assert len(st.mesh_refinement_iter) >= 1          # the mesh followed the moving switching times
        ts = solver.event_times[0]
        assert np.abs(ts - ts0[0]).min() > 0.02            # both switching times were really optimised
        dwell = np.diff(np.concatenate([[0.0], ts, [info["T"]]]))
        assert (dwell > np.array(solver.sto.minimum_dwell_times) - 1e-9).all()   # ... inside their minimum dwell times
        rows = solver.ctx.sto_constraint_data()[0]
        assert (rows[0] > 0).all() and (rows[1] > 0).all()
        assert np.abs(rows[0] * rows[1] - solver.sto.barrier_param).max() < 1e-6   # central path of the dwell-time rows
        # the time steps on the device are those of the optimised event times, per phase
        sol = solver.get_solution()[0]
        worst, t, dt = _residuals(oracle, solver, 0, sol)
        ev = [i for i, g in enumerate(solver.grids[:-1]) if g.type in (1, 2)]
        # (the table on the device belongs to the iterate the last iteration linearised at: one converged step behind `ts`)
        assert np.abs(t[ev] - ts).max() < 1e-9
        print("converged jump, worst residuals by the CPU restatement:", worst)
        assert worst["IDC"] < 1e-7 and worst["impact"] < 1e-7 and worst["Fx"] < 1e-8 and worst["switching"] < 1e-8
        # it really jumps: the feet are off the ground in the flight phase, and the base ends up further ahead
        S = Records(solver.ctx.L, "sol")
        flight = [i for i, mk in enumerate(solver.masks[:-1]) if mk == 0]
        zmax = max(oracle.rbd_contact_position(solver.model, S.f(sol[i], "q")[:solver.model.nq], 0)[2] for i in flight)
        assert zmax > 0.01
        assert S.f(sol[-1], "q")[0] > x0[0, 0] + 0.1
        # OCPSolver::KKTError(t, q, v) at the converged iterate, without moving it
        assert solver.kkt_error()[0] < 1e-6
    finally:
        solver.close()


@pytest.mark.gpu
def test_batched_jump_instances_optimise_their_own_switching_times(oracle):
    """Three instances from different initial states share the grid structure, not the event times: each converges to its own
    switching times and time steps."""
    from robotoc_amd import problems_jump as pj
    solver, x0, info = pj.anymal_jump_sto_solver(batch=3, x0_noise=0.05)
    try:
        st = solver.solve(0.0, x0)
        hist = np.array([e.max() for e in st.kkt_error])
        print("3 jumps: %d iterations, worst KKT %.1e, event times\n%s" % (st.iter, hist[-1], solver.event_times))
        assert st.convergence and (solver.ctx.status() == 0).all()
        ts = solver.event_times
        assert np.abs(ts[0] - ts[1]).max() > 1e-5 and np.abs(ts[0] - ts[2]).max() > 1e-5
        dts = solver.ctx.sto_time_steps()
        assert np.abs(dts[0] - dts[1]).max() > 1e-7
        sol = solver.get_solution()
        for b in range(3):
            worst, _, _ = _residuals(oracle, solver, b, sol[b])
            assert worst["IDC"] < 1e-7 and worst["Fx"] < 1e-8 and worst["switching"] < 1e-8, (b, worst)
    finally:
        solver.close()


@pytest.mark.gpu
def test_icub_jump_example_converges_on_the_device():
    """BASELINE configs[3] as the reference poses it (examples/icub/python/jump_sto.py): iCub, nv = 35, two SURFACE contacts, two jumps of
    0.5 m, all four switching times optimised, the example's ConfigurationSpaceCost weights, the joint limits of its URDF, FrictionCone on
    the soles (mu = 0.6), minimum dwell times [0.6, 0.2, 0.6, 0.2, 0.6], T = 2.6 s, N = 130, kkt_tol_mesh = 0.1, max_dt_mesh = T / N,
    initial_sto_reg_iter = 10, max_iter = 350 -- OCPSolver::solve from the example's initial guess (the standing pose on every grid
    point) converges to its kkt_tol with two mesh refinements; every iteration runs on the device."""
    from robotoc_amd import problems_jump as pj
    solver, x0, info = pj.icub_jump_sto_solver(batch=1)
    try:
        assert info["N"] == 130 and solver.dims.nv == 35 and solver.cone_dim == 6
        st = solver.solve(0.0, x0)
        errs = np.array([float(np.max(e)) for e in st.kkt_error])
        ts = solver.event_times[0]
        print("iCub jump_sto example: %d iterations, converged %s, KKT %.2e -> %.2e, mesh refinements at %s, event times %s"
              % (st.iter, st.convergence, errs[0], errs[-1], st.mesh_refinement_iter, np.array2string(ts, precision=4)))
        assert st.convergence and st.iter < 350 and errs[-1] < 1e-7 and errs[0] > 1e3
        assert (solver.ctx.status() == 0).all()
        assert len(st.mesh_refinement_iter) >= 1
        # the optimised switching times respect the minimum dwell times and moved off the initial ones
        gaps = np.diff(np.concatenate([[0.0], ts, [info["T"]]]))
        assert (gaps >= np.array(info["min_dwell"]) - 1e-9).all()
        assert np.abs(ts - np.array([0.7, 0.95, 1.65, 1.9])).max() > 0.02
        # the solution is dynamically consistent: the base travels the two jump lengths
        S = Records(solver.ctx.L, "sol")
        q = S.f(solver.get_solution()[0], "q")
        assert abs(q[len(solver.grids) - 1, 0] - 1.0) < 0.15
    finally:
        solver.close()
