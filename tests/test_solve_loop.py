"""rtoc_solve_loop (include/rtoc_robot.h): the iteration schedule of OCPSolver::solve (reference src/solver/ocp_solver.cpp:169-213) that both
host shells run -- robotoc_amd/solver.py through robotoc_amd.capi.solve_loop, robotoc::OCPSolver::solve of
robotoc_amd/host/robotoc_hip_solver.hpp directly.  Pure host logic: checked here without a GPU against a line-by-line restatement of the
reference's loop, on scripted sequences of KKT errors and time steps (the trace of calls and their arguments has to be the same)."""
import numpy as np
import pytest

from robotoc_amd import capi


def reference_loop(max_iter, kkt_tol, sto, reg_iter, reg, tol_mesh, max_dt_mesh, errors, max_dts):
    """ocp_solver.cpp:169-213 with the members replaced by a trace"""
    trace, conv, it_out, refined = [], False, None, []
    inner_iter, iter_ = 0, 0
    k_err, k_dt = 0, 0
    while iter_ < max_iter:                                            # for (int iter=0; iter<max_iter; ++iter, ++inner_iter)
        if sto:                                                        # :171-177
            trace.append(("reg", reg if inner_iter < reg_iter else 0.0))
        trace.append(("update",))                                      # :178
        kkt_error = errors[k_err]; k_err += 1                          # :180
        if sto and kkt_error < tol_mesh:                               # :181
            dt = max_dts[k_dt]; k_dt += 1
            trace.append(("max_dt",))
            if dt > max_dt_mesh:                                       # :182-199
                trace.append(("refine",))
                inner_iter = 0
                refined.append(iter_ + 1)
            elif kkt_error < kkt_tol:                                  # :200-204
                conv, it_out = True, iter_ + 1
                break
        elif kkt_error < kkt_tol:                                      # :206-210
            conv, it_out = True, iter_ + 1
            break
        iter_ += 1
        inner_iter += 1
    if not conv:                                                       # :212-214
        it_out = max_iter
    return trace, conv, it_out, refined


def run_library(max_iter, kkt_tol, sto, reg_iter, reg, tol_mesh, max_dt_mesh, errors, max_dts):
    trace = []
    e, d = iter(errors), iter(max_dts)

    def update():
        trace.append(("update",))
        return next(e)

    def max_dt():
        trace.append(("max_dt",))
        return next(d)
    conv, it, refined = capi.solve_loop(max_iter, kkt_tol, update, sto_enabled=sto, initial_sto_reg_iter=reg_iter, initial_sto_reg=reg,
                                        kkt_tol_mesh=tol_mesh, max_dt_mesh=max_dt_mesh,
                                        set_sto_regularization=(lambda r: trace.append(("reg", r))) if sto else None,
                                        max_time_step=max_dt if sto else None,
                                        mesh_refinement=(lambda: trace.append(("refine",))) if sto else None)
    return trace, conv, it, refined


@pytest.mark.parametrize("seed", range(40))
def test_schedule_is_the_reference_loop(seed):
    rng = np.random.default_rng(seed)
    max_iter = int(rng.integers(0, 25))
    sto = bool(seed % 2)
    kkt_tol, tol_mesh = 1e-3, float(rng.choice([1e-3, 0.1, 10.0]))
    reg_iter, reg = int(rng.integers(0, 5)), 1.0e30
    max_dt_mesh = 0.02
    errors = list(10.0 ** rng.uniform(-5, 1.5, 64))
    if seed % 3 == 0:
        errors = sorted(errors, reverse=True)         # a converging run
    max_dts = list(rng.choice([0.01, 0.03], 64))
    args = (max_iter, kkt_tol, sto, reg_iter, reg, tol_mesh, max_dt_mesh, errors, max_dts)
    want, got = reference_loop(*args), run_library(*args)
    assert got[0] == want[0], (got[0][:12], want[0][:12])
    assert got[1:] == want[1:]


def test_regularisation_restarts_at_one_after_a_refinement():
    """`inner_iter = 0` inside the body, then the loop header's ++inner_iter (ocp_solver.cpp:169, :197): the iteration behind a mesh
    refinement runs with inner_iter = 1 -- with initial_sto_reg_iter = 1 it is NOT regularised again."""
    errors = [1.0, 0.05, 0.05, 1e-9]
    trace, conv, it, refined = run_library(10, 1e-6, True, 1, 1.0e30, 0.1, 0.02, errors, [0.03, 0.01, 0.01])
    regs = [x[1] for x in trace if x[0] == "reg"]
    assert regs == [1.0e30, 0.0, 0.0, 0.0] and refined == [2] and conv and it == 4


def test_without_convergence_iter_is_max_iter_and_errors_propagate():
    trace, conv, it, refined = run_library(5, 1e-9, False, 0, 0.0, 0.1, 0.0, [1.0] * 8, [])
    assert not conv and it == 5 and len(trace) == 5 and refined == []

    def boom():
        raise ValueError("from the update callback")
    with pytest.raises(ValueError, match="from the update callback"):
        capi.solve_loop(3, 1e-9, boom)
    with pytest.raises(capi.RtocError):      # an STO problem needs its callbacks
        capi.solve_loop(3, 1e-9, lambda: 1.0, sto_enabled=True)
