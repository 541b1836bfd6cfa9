"""The ANYmal jump with switching-time optimisation of the reference's examples/anymal/python/jump_sto.py (BASELINE configs[2]),
scaled to the N = 40 horizon BASELINE.json names: stand -> flight -> stand, lift-off and touch-down times optimised
(ContactSequence::push_back(..., sto=True), :96-103), ConfigurationSpaceCost with the example's weights (:25-57), its Constraints
object (six joint-limit components + FrictionCone, :60-74), STOConstraints with minimum dwell times (:107-109), the example's
solver options (kkt_tol_mesh = 1, max_dt_mesh = T / N, :117-122).  Everything the example sets up with Pinocchio on the host
(foot positions at the standing pose, total weight) comes from robotoc_amd.robot_model."""
import numpy as np

from . import robot_model as rm
from .grid import ANYMAL_Q_STANDING, Event
from .solver import ContactPlan, OCPSolver, SolverOptions, STOConstraints
from .types import Records


def anymal_jump_sto_solver(batch=1, device=0, N=40, dt=0.02, jump_length=0.25, ground_time=0.31, flying_time=0.2,
                           min_dwell=(0.1, 0.1, 0.2), with_limits=True, with_cones=True, max_iter=200, seed=7, x0_noise=0.0,
                           horizon_scan="off"):
    m = rm.load_named("anymal")
    nv, nq, nu = m.nv, m.nq, m.nu
    qs = np.array(ANYMAL_Q_STANDING, dtype=float)
    T = N * dt
    feet = np.array([m.frame_placement(qs, c)[1] for c in range(4)])
    landed = feet + np.array([jump_length, 0.0, 0.0])
    plan = ContactPlan([0b1111, 0, 0b1111], [feet, feet, landed],
                       [Event("lift", ground_time, sto=True), Event("impact", ground_time + flying_time, sto=True)])
    q_ref = qs.copy()
    q_ref[0] += jump_length
    wq = np.concatenate([[1.0, 0.0, 0.0, 1.0, 1.0, 1.0], np.full(12, 0.001)])
    wq_imp = np.concatenate([[0.0, 0.0, 0.0, 100.0, 100.0, 100.0], np.full(12, 0.1)])
    cost = dict(q_ref=q_ref, v_ref=np.zeros(nv), u_ref=np.zeros(nu), q_weight=wq, v_weight=np.full(nv, 1.0), a_weight=np.full(nv, 1e-6),
                u_weight=np.zeros(nu), q_weight_terminal=wq, v_weight_terminal=np.full(nv, 1.0), q_weight_impact=wq_imp,
                v_weight_impact=np.full(nv, 1.0), dv_weight_impact=np.full(nv, 1e-6))
    # joint limits of the ANYmal URDF (anymal_b_simple_description): position +-9.42 (continuous joints), velocity 7.5 rad/s, effort 80 N m
    limits = (np.full(nu, -9.42), np.full(nu, 9.42), np.full(nu, 7.5), np.full(nu, 80.0)) if with_limits else None
    opts = SolverOptions(max_iter=max_iter, kkt_tol=1e-7, kkt_tol_mesh=1.0, max_dt_mesh=T / N, horizon_scan=horizon_scan)
    solver = OCPSolver(m, plan, T, N, cost, joint_limits=limits, friction_coefficients=np.full(4, 0.7) if with_cones else None,
                       sto_constraints=STOConstraints(list(min_dwell)), options=opts, batch=batch, device=device)
    rng = np.random.default_rng(seed)
    x0 = np.tile(np.concatenate([qs, np.zeros(nv)]), (batch, 1))
    if x0_noise > 0.0:
        x0[:, 7:nq] += x0_noise * rng.uniform(-1, 1, (batch, nq - 7))
        x0[:, nq:] = x0_noise * rng.uniform(-1, 1, (batch, nv))
    solver.discretize(0.0)
    # initial guess like the example: q, v = the initial state on every grid point, f = a quarter of the weight per foot (:133-137)
    S = Records(solver.ctx.L, "sol")
    sol = S.zeros(batch, len(solver.grids))
    S.f(sol, "q")[..., :nq] = x0[:, None, :nq]
    S.f(sol, "v")[...] = x0[:, None, nq:]
    weight = 9.81 * sum(m.mass[i] for i in range(m.njoints))
    for i, g in enumerate(solver.grids):
        act = [k for k in range(4) if (int(solver.masks[i]) >> k) & 1]
        if act and g.type != 1 and i < len(solver.grids) - 1:
            S.f(sol, "f")[:, i, :3 * len(act)] = np.tile([0.0, 0.0, 0.25 * weight], len(act))
    solver.set_solution(sol)
    return solver, x0, dict(T=T, N=N, model=m, cost=cost, min_dwell=list(min_dwell),
                            limits=[9.42, 7.5, 80.0, 0.7])
