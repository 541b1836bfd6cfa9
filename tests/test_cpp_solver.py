"""robotoc::OCPSolver / robotoc::DirectMultipleShooting shells (robotoc_amd/host/robotoc_hip_solver.hpp) on the GPU:
Python records a stage dump of an ANYmal trot problem at the evalKKT boundary (pre-condensation records, the 72
joint-limit rows, the friction cones, initial state direction and iterate), the C++ program
tests/cpp/ocp_solver_test.cpp runs OCPSolver::updateSolution on it through the shells, and its outputs are compared
with the CPU oracle's sequence of the same iteration (ocp_solver.cpp:111-145 downstream of the linearisation)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import rel_err
from robotoc_amd import problems as pr
from robotoc_amd.types import (BUF_CDD, BUF_CON, BUF_CONE, BUF_DX0, BUF_KKT, BUF_SOL, Records, joint_limit_rows)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MC, CD = 4, 3


@pytest.mark.gpu
@pytest.mark.parametrize("scan", [False])
def test_ocp_solver_update_solution_matches_the_oracle(tmp_path, oracle, scan):
    from robotoc_amd import capi
    from test_cpp_host import _build
    exe = _build("ocp_solver_test")
    dims, grids, _ = pr.config_anymal_trot()
    n = len(grids)
    ctx = capi.Context(dims, n, 1, 0)
    L = ctx.L
    ctx.set_grid(grids)
    rows = joint_limit_rows(dims)
    ctx.set_constraint_rows(rows)
    ctx.set_friction_cones(MC, CD)
    kkt, cdd = pr.make_precondense_batch(L, grids, 1)
    con = pr.make_constraint_batch(L, grids, 1)
    cone = pr.make_cone_batch(L, grids, 1, MC)
    dx0 = pr.make_dx0(L, 1)
    sol = np.random.default_rng(5).uniform(-1, 1, (1, n, L.sol.stride))
    for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, con), (BUF_CONE, cone), (BUF_DX0, dx0), (BUF_SOL, sol)):
        ctx.upload(buf, arr)
    dump = str(tmp_path / "anymal_trot.rtocdump")
    ctx.save_stage_dump(dump, (BUF_KKT, BUF_CDD, BUF_CON, BUF_CONE, BUF_DX0, BUF_SOL))
    ctx.close()
    out_path = str(tmp_path / "solver_out.bin")
    run = subprocess.run([exe, dump, out_path], capture_output=True, text=True, timeout=300)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    raw = np.fromfile(out_path)
    kkt_error, primal, dual, nn = raw[:4]
    assert int(nn) == n
    nv, nu, nx, nvf = dims.nv, dims.nu, 2 * dims.nv, dims.nv + dims.nf_max
    sizes = [("dx", nx), ("du", nu), ("dlmdgmm", nx), ("daf", nvf), ("dbetamu", nvf), ("q", nv + 1), ("v", nv), ("a", nv), ("u", nu),
             ("lmd", nv), ("gmm", nv), ("s", nx), ("k", nu)]
    per = sum(sz for _, sz in sizes)
    body = raw[4:].reshape(n, per)
    got, o = {}, 0
    for name, sz in sizes:
        got[name] = body[:, o:o + sz]
        o += sz
    # ---- the oracle's sequence of the same iteration ----
    tau = 0.995
    err_ref = oracle.kkt_error(L, grids, kkt, cdd, con, rows, MC, CD, 5)[0]
    assert abs(kkt_error - err_ref) <= 1e-12 * err_ref
    kk, cc, nn_ = kkt.copy(), cdd.copy(), con.copy()
    oracle.pdipm_condense_batch(L, grids, rows, kk, nn_)
    oracle.cone_condense_batch(L, grids, MC, CD, cone, kk, cc, nn_)
    assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
    R, D, S = Records(L, "ric"), Records(L, "dir"), Records(L, "sol")
    ric_ref, d_ref = R.zeros(1, n), D.zeros(1, n)
    oracle.riccati_sweep_batch(L, grids, kk, ric_ref, d_ref, dx0=dx0)
    oracle.expand_batch(L, grids, cc, d_ref)
    steps_ref = oracle.pdipm_expand_batch(L, grids, rows, nn_, d_ref, tau)
    oracle.cone_expand_batch(L, grids, MC, CD, cone, nn_, d_ref, tau, steps_ref)
    assert np.allclose([primal, dual], steps_ref[0], rtol=1e-6), ((primal, dual), steps_ref)
    worst = 0.0
    for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu"):
        e = rel_err(got[f], D.f(d_ref[0], f))
        worst = max(worst, e)
        assert e < 1e-7, (f, e)
    assert rel_err(got["s"], R.f(ric_ref[0], "s")) < 1e-7
    assert rel_err(got["k"][:-1], R.f(ric_ref[0], "k")[:-1]) < 1e-7
    sol_ref = sol.copy()
    oracle.integrate_solution_batch(L, grids, np.array([[primal, dual]]), d_ref, sol_ref)
    for f in ("q", "v", "a", "u", "lmd", "gmm"):
        e = rel_err(got[f], S.f(sol_ref[0], f))
        worst = max(worst, e)
        assert e < 1e-8, (f, e)
    print("OCPSolver::updateSolution (C++ shell) vs oracle sequence: worst rel err %.3e" % worst)


@pytest.mark.gpu
def test_unconstr_ocp_solver_solves_iiwa14_on_the_device(tmp_path):
    """robotoc::UnconstrOCPSolver::solve (C++ mirror) = the same iterations issued through ctypes, and it converges."""
    import ctypes as C
    from robotoc_amd import capi, robot_model as rm
    from robotoc_amd.robot_model import MAX_JOINTS
    from test_cpp_host import _build
    exe = _build("unconstr_ocp_solver_test")
    dims, grids, meta = pr.config_iiwa14()
    m = rm.load_named("iiwa14")
    n, nv, dt = len(grids), m.nv, meta["dt"]
    rng = np.random.default_rng(21)
    cost = np.zeros((12, MAX_JOINTS))
    cost[0, :nv] = rng.uniform(-0.8, 0.8, nv)
    for k, w in ((3, 10.0), (4, 0.1), (5, 0.01), (6, 0.001), (7, 10.0), (8, 0.1)):
        cost[k, :nv] = w
    q0, v0 = rng.uniform(-0.5, 0.5, nv), np.zeros(nv)
    prob = str(tmp_path / "iiwa14_problem.bin")
    with open(prob, "wb") as f:
        f.write(bytes(m))
        f.write(cost.tobytes())
        f.write(np.array([dt * (n - 1)]).tobytes())
        f.write(np.array([n - 1], dtype=np.int32).tobytes())
        f.write(q0.tobytes())
        f.write(v0.tobytes())
    out_path = str(tmp_path / "unconstr_out.bin")
    run = subprocess.run([exe, prob, out_path], capture_output=True, text=True, timeout=300)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    raw = np.fromfile(out_path)
    iters, conv, err, err0 = int(raw[0]), raw[1], raw[2], raw[3]
    assert conv == 1.0 and err < 1e-9 and iters <= 20
    traj = raw[4:4 + n * 2 * nv].reshape(n, 2 * nv)
    # the same iterations through ctypes
    ctx = capi.Context(dims, n, 1, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_configuration_cost(*[cost[k, :nv] for k in range(9)])
    ctx.set_initial_state(np.concatenate([q0, v0])[None])
    S = Records(ctx.L, "sol")
    sol = S.zeros(1, n)
    S.f(sol, "q")[..., :nv] = q0
    ctx.upload(BUF_SOL, sol)
    errs = [ctx.unconstr_update_solution(dt)[0] for _ in range(iters)]
    assert abs(errs[0] - err0) <= 1e-12 * err0 and abs(errs[-1] - err) <= 1e-6 * max(err, 1e-12)
    sol = ctx.download_records(BUF_SOL, "sol")
    ref = np.concatenate([S.f(sol[0], "q")[:, :nv], S.f(sol[0], "v")], axis=1)
    assert np.array_equal(traj, ref)
    ctx.close()


@pytest.mark.gpu
def test_ocp_solver_with_device_linearisation_solves_anymal_standing(tmp_path, oracle):
    """robotoc::OCPSolver::solve through the ConfigurationCostSource (robotoc_hip_device_source.hpp): the whole iteration on
    the device; the trajectory equals the ctypes loop of rtoc_contact_update_solution bit for bit."""
    from robotoc_amd import capi, robot_model as rm
    from robotoc_amd.grid import uniform_grid
    from robotoc_amd.robot_model import MAX_JOINTS
    from robotoc_amd.types import anymal_dims
    from test_cpp_host import _build
    from test_contact_closed_loop import Q_STAND
    exe = _build("ocp_solver_device_test")
    m = rm.load_named("anymal")
    N, dt = 20, 0.02
    nv, nq = m.nv, m.nq
    feet = np.array([oracle.rbd_contact_position(m, Q_STAND, c) for c in range(4)])
    q_ref = Q_STAND.copy()
    q_ref[:7] = oracle.se3_integrate(Q_STAND[:7], np.array([0.03, 0.0, -0.02, 0.0, 0.05, 0.0]))
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 0.1)])
    cost = np.zeros((12, MAX_JOINTS))
    for k, val in ((0, q_ref), (3, wq), (4, np.full(nv, 1.0)), (5, np.full(nv, 1e-3)), (6, np.full(12, 1e-3)), (7, 10.0 * wq), (8, np.full(nv, 1.0))):
        cost[k, :len(val)] = val
    q0 = Q_STAND.copy()
    q0[:7] = oracle.se3_integrate(Q_STAND[:7], np.array([0.005, -0.004, 0.003, 0.01, -0.006, 0.004]))
    v0 = np.zeros(nv)
    mass = sum(m.mass[i] for i in range(m.njoints))
    f0 = np.concatenate([oracle.rbd_contact_placement(m, q0, c)[0].T @ np.array([0.0, 0.0, 9.81 * mass / 4]) for c in range(4)])
    u0 = oracle.rbd_eval(m, 0, q0, np.zeros(nv), np.zeros(nv), f0, np.zeros(12), 0b1111, feet.reshape(-1))[6:nv]
    prob = str(tmp_path / "anymal_standing.bin")
    with open(prob, "wb") as f:
        f.write(bytes(m))
        f.write(cost.tobytes())
        f.write(np.array([N], dtype=np.int32).tobytes())
        f.write(np.array([dt]).tobytes())
        for arr in (q0, v0, feet.reshape(-1), f0, u0):
            f.write(np.ascontiguousarray(arr, dtype=np.float64).tobytes())
    out_path = str(tmp_path / "ocp_device_out.bin")
    run = subprocess.run([exe, prob, out_path], capture_output=True, text=True, timeout=300)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    raw = np.fromfile(out_path)
    iters, conv, err = int(raw[0]), raw[1], raw[2]
    assert conv == 1.0 and err < 1e-8 and iters <= 20
    traj = raw[4:].reshape(N + 1, nq)
    # the same iterations through ctypes
    grids = uniform_grid(N, dt, dimf=12)
    ctx = capi.Context(anymal_dims(), N + 1, 1, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_contact_schedule(np.full(N + 1, 0b1111, dtype=np.uint32), np.tile(feet[None], (N + 1, 1, 1)))
    ctx.set_configuration_cost(*[cost[k, :(nq if k == 0 else nv)] for k in range(12)])
    ctx.set_initial_state(np.concatenate([q0, v0])[None])
    S = Records(ctx.L, "sol")
    sol = S.zeros(1, N + 1)
    S.f(sol, "q")[..., :nq] = q0
    S.f(sol, "f")[..., :12] = f0
    S.f(sol, "u")[..., :12] = u0
    ctx.upload(BUF_SOL, sol)
    for _ in range(iters):
        ctx.contact_update_solution()
    ref = S.f(ctx.download_records(BUF_SOL, "sol")[0], "q")[:, :nq]
    assert np.array_equal(traj, ref)
    ctx.close()


@pytest.mark.gpu
def test_ocp_solver_constrained_trot_on_the_device(tmp_path, oracle):
    """robotoc::OCPSolver::solve over ConfigurationCostSource with joint limits and friction cones on BASELINE configs[1]'s
    contact sequence (47 grid points: lifts, touch-downs with switching constraints): every part of the iteration on the
    device, and bit-identical to the same launch sequence driven through ctypes."""
    from robotoc_amd import capi, robot_model as rm
    from robotoc_amd.problems import config_anymal_trot
    from robotoc_amd.robot_model import MAX_JOINTS
    from robotoc_amd.types import GRID_IMPACT, joint_limit_rows
    from test_cpp_host import _build
    from test_contact_closed_loop import Q_STAND
    from test_contact_constraints import limits
    from test_switching_constraint_lin import trot_masks
    exe = _build("ocp_solver_trot_test")
    m = rm.load_named("anymal")
    dims, grids, _ = config_anymal_trot()
    n, nv, nq, nu = len(grids), m.nv, m.nq, 12
    masks = trot_masks(grids, [0b1111, 0b1001, 0b1111, 0b0110, 0b1111], [0b0110, 0b1001])
    feet = np.array([oracle.rbd_contact_position(m, Q_STAND, c) for c in range(4)])
    pos = np.tile(feet[None], (n, 1, 1))
    impacts = [i for i, g in enumerate(grids) if g.type == GRID_IMPACT]
    pos[impacts[0]:, [1, 2], 0] += 0.05
    pos[impacts[1]:, [0, 3], 0] += 0.05
    q_ref = Q_STAND.copy()
    q_ref[0] += 0.15
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])
    cost = np.zeros((12, MAX_JOINTS))
    for k, val in ((0, q_ref), (3, wq), (4, np.full(nv, 1.0)), (5, np.full(nv, 1e-3)), (6, np.full(12, 1e-3)), (7, 10.0 * wq), (8, np.full(nv, 1.0)),
                   (9, wq), (10, np.full(nv, 1.0)), (11, np.full(nv, 1e-3))):
        cost[k, :len(val)] = val
    q0, v0 = Q_STAND.copy(), np.zeros(nv)
    mass = sum(m.mass[i] for i in range(m.njoints))
    finit = np.zeros((n, 12))
    for i in range(n):
        act = [c for c in range(4) if (int(masks[i]) >> c) & 1]
        if act and grids[i].type != GRID_IMPACT:
            finit[i, :3 * len(act)] = np.concatenate([oracle.rbd_contact_placement(m, q0, c)[0].T @ np.array([0.0, 0.0, 9.81 * mass / len(act)]) for c in act])
    qmax, vmax, umax, mu, barrier = 1.2, 3.0, 17.34, 0.2, 1.0e-3
    prob = str(tmp_path / "anymal_trot.bin")
    with open(prob, "wb") as f:
        f.write(bytes(m))
        f.write(cost.tobytes())
        f.write(np.array([n], dtype=np.int32).tobytes())
        for g in grids:
            f.write(bytes(g))
        f.write(masks.astype(np.uint32).tobytes())
        for arr in (pos, q0, v0, finit, np.array([qmax, vmax, umax, mu, barrier])):
            f.write(np.ascontiguousarray(arr, dtype=np.float64).tobytes())
    out_path = str(tmp_path / "ocp_trot_out.bin")
    run = subprocess.run([exe, prob, out_path], capture_output=True, text=True, timeout=300)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    raw = np.fromfile(out_path)
    iters, conv, err = int(raw[0]), raw[1], raw[2]
    assert conv == 1.0 and err < 1e-8 and iters <= 40
    traj = raw[4:4 + n * nq].reshape(n, nq)
    torque = raw[4 + n * nq:].reshape(n, nu)
    assert np.abs(torque).max() < umax
    # the same iterations through ctypes
    ctx = capi.Context(dims, n, 1, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_contact_schedule(masks, pos)
    ctx.set_constraint_rows(joint_limit_rows(dims))
    ctx.set_friction_cones(4, 3)
    ctx.set_impact_cones(False)
    ctx.set_constraint_bounds(limits(nu, qmax, vmax, umax), barrier, 0.995)
    ctx.set_friction_coefficients(np.full(4, mu))
    ctx.set_configuration_cost(*[cost[k, :(nq if k == 0 else nv)] for k in range(12)])
    ctx.set_initial_state(np.concatenate([q0, v0])[None])
    S = Records(ctx.L, "sol")
    sol = S.zeros(1, n)
    S.f(sol, "q")[..., :nq] = q0
    S.f(sol, "f")[0, :, :12] = finit
    ctx.upload(BUF_SOL, sol)
    ctx.contact_init_constraints()
    for _ in range(iters):
        ctx.contact_update_solution(0.995)
    ref = ctx.download_records(BUF_SOL, "sol")[0]
    assert np.array_equal(traj, S.f(ref, "q")[:, :nq]) and np.array_equal(torque, S.f(ref, "u")[:, :nu])
    ctx.close()


@pytest.mark.gpu
def test_ocp_solver_jump_with_switching_time_optimisation_on_the_device(tmp_path):
    """BASELINE configs[2] through the C++ shell: robotoc::OCPSolver::solve of the ANYmal jump with two STO-enabled events --
    SwitchingTimeOptimization mirror, regularisation schedule, mesh refinement with solution interpolation in C++
    (robotoc_hip_planner.hpp) -- against the same solve by the Python mirror (robotoc_amd/solver.py): same number of iterations,
    same mesh refinement, the same optimised switching times."""
    from robotoc_amd import problems_jump as pj
    from robotoc_amd.robot_model import MAX_JOINTS
    from test_cpp_host import _build
    exe = _build("ocp_solver_jump_sto_test")
    solver, x0, info = pj.anymal_jump_sto_solver(batch=1)
    m = info["model"]
    nv, nq, nu = m.nv, m.nq, m.nu
    try:
        st = solver.solve(0.0, x0)
        assert st.convergence
        hist = np.array([e[0] for e in st.kkt_error])
        ts_py = solver.event_times[0].copy()
        plan, opts = solver.plan, solver.options
    finally:
        solver.close()
    cost = np.zeros((12, MAX_JOINTS))
    c = info["cost"]
    for k, key in enumerate(("q_ref", "v_ref", "u_ref", "q_weight", "v_weight", "a_weight", "u_weight", "q_weight_terminal", "v_weight_terminal",
                             "q_weight_impact", "v_weight_impact", "dv_weight_impact")):
        cost[k, :len(c[key])] = c[key]
    weight = 9.81 * sum(m.mass[i] for i in range(m.njoints))
    prob = str(tmp_path / "anymal_jump_sto.bin")
    with open(prob, "wb") as f:
        f.write(bytes(m))
        f.write(cost.tobytes())
        f.write(np.array([info["N"]], dtype=np.int32).tobytes())
        for arr in ([info["T"], plan.events[0].time, plan.events[1].time], plan.phase_positions[0], plan.phase_positions[2], x0[0, :nq], x0[0, nq:],
                    np.tile([0.0, 0.0, 0.25 * weight], 4), info["min_dwell"], info["limits"], [opts.max_dt_mesh]):
            f.write(np.ascontiguousarray(arr, dtype=np.float64).tobytes())
    out_path = str(tmp_path / "jump_sto_out.bin")
    run = subprocess.run([exe, prob, out_path], capture_output=True, text=True, timeout=300)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    raw = np.fromfile(out_path)
    iters, conv, err, nref, first_ref, ts_cpp, hist_cpp = int(raw[0]), raw[1], raw[2], int(raw[3]), int(raw[4]), raw[5:7], raw[7:]
    assert conv == 1.0 and err < 1e-7
    assert nref == len(st.mesh_refinement_iter) >= 1 and first_ref == st.mesh_refinement_iter[0]
    # up to the first mesh refinement both shells issue the same launches on the same data: the iterates are bit-identical, and so
    # are the KKT errors up to the LAST rounding of how each shell assembles them -- the Python shell takes sqrt(dms + sto) from the
    # device, the C++ shell squares the device's sqrt(dms) again before it adds the STO term (PerformanceIndex keeps squares): one ulp
    # at some iterations, none at the next (round 6: iterations 5 and 7 of 12; that the neighbours agree bit for bit IS the evidence
    # that the iterates do).  Behind the refinement the interpolated warm start agrees to rounding, so the paths stay together.
    rel = np.abs(hist_cpp[:first_ref] - hist[:first_ref]) / hist[:first_ref]
    assert rel.max() <= 4 * np.finfo(float).eps and (rel == 0.0).sum() >= first_ref // 2, rel
    assert abs(iters - st.iter) <= 1
    assert np.abs(ts_cpp - ts_py).max() < 1e-6, (ts_cpp, ts_py)


@pytest.mark.gpu
def test_ocp_solver_icub_jump_example_through_the_cpp_shell(tmp_path):
    """BASELINE configs[3] as the reference poses it (examples/icub/python/jump_sto.py) through the C++ shell: a robotoc::ContactSequence
    of SURFACE contacts with their placements (positions + rotations), four STO-enabled events, the example's cost, limits, FrictionCone,
    STOConstraints and solver options; robotoc::OCPSolver::solve converges like the Python mirror does -- same iteration count (+-1),
    the same mesh refinements, the same optimised switching times."""
    from robotoc_amd import problems_jump as pj
    from robotoc_amd.robot_model import MAX_JOINTS
    from test_cpp_host import _build
    exe = _build("ocp_solver_icub_jump_sto_test")
    solver, x0, info = pj.icub_jump_sto_solver(batch=1)
    m = info["model"]
    nv, nq, nu = m.nv, m.nq, m.nu
    try:
        st = solver.solve(0.0, x0)
        assert st.convergence
        hist = np.array([e[0] for e in st.kkt_error])
        ts_py = solver.event_times[0].copy()
        plan, opts = solver.plan, solver.options
    finally:
        solver.close()
    cost = np.zeros((12, MAX_JOINTS))
    c = info["cost"]
    for k, key in enumerate(("q_ref", "v_ref", "u_ref", "q_weight", "v_weight", "a_weight", "u_weight", "q_weight_terminal", "v_weight_terminal",
                             "q_weight_impact", "v_weight_impact", "dv_weight_impact")):
        cost[k, :len(c[key])] = c[key]
    nev = len(plan.events)
    prob = str(tmp_path / "icub_jump_sto.bin")
    with open(prob, "wb") as f:
        f.write(bytes(m))
        f.write(cost.tobytes())
        f.write(np.array([info["N"], nev], dtype=np.int32).tobytes())
        for arr in ([info["T"]], [e.time for e in plan.events], np.array(plan.phase_positions), plan.phase_rotations[0], x0[0, :nq], x0[0, nq:],
                    info["min_dwell"][:nev + 1], pj.ICUB_Q_MIN, pj.ICUB_Q_MAX, pj.ICUB_V_MAX, pj.ICUB_U_MAX, [0.6], [opts.max_dt_mesh]):
            f.write(np.ascontiguousarray(arr, dtype=np.float64).tobytes())
    out_path = str(tmp_path / "icub_jump_sto_out.bin")
    run = subprocess.run([exe, prob, out_path], capture_output=True, text=True, timeout=600)
    print(run.stdout, run.stderr)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    raw = np.fromfile(out_path)
    iters, conv, err, nref, first_ref, ts_cpp, hist_cpp = int(raw[0]), raw[1], raw[2], int(raw[3]), int(raw[4]), raw[5:5 + nev], raw[5 + nev:]
    assert conv == 1.0 and err < 1e-7
    assert nref == len(st.mesh_refinement_iter) >= 1 and abs(first_ref - st.mesh_refinement_iter[0]) <= 1
    assert np.allclose(hist_cpp[:10], hist[:10], rtol=1e-9)   # the same launches on the same data up to the end of the regularised iterations
    assert abs(iters - st.iter) <= 3
    assert np.abs(ts_cpp - ts_py).max() < 1e-5, (ts_cpp, ts_py)
