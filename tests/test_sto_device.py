"""The switching-time half of OCPSolver::updateSolution on the device (robotoc_amd/csrc/sto.hpp):
TimeDiscretization::correctTimeSteps, STOConstraints (minimum dwell times as PDIPM rows), SwitchingTimeOptimization::evalKKT /
computeStepSizes / integrateSolution -- per instance, the batch sharing only the grid STRUCTURE.

CPU: oracle/sto.py (the numpy restatement the kernels are held to) against the reference's own sources -- its TimeDiscretization
through robotoc_amd/grid.py (identical to it: tests/test_discretization_and_filter_vs_reference.py), its
SwitchingTimeOptimization::evalKKT (oracle/_ref: ref_sto_eval_kkt), and the STO iteration fixture written by its
SwitchingTimeOptimization inside a whole OCPSolver::updateSolution (tests/golden/ref_anymal_jump_sto_solver_iteration.npz).
GPU: every rtoc_sto_* entry point against oracle/sto.py on batches with a different set of event times per instance; the whole
iteration against the fixture is tests/test_golden_ref.py::test_ocp_solver_iteration_replays_the_reference_sources."""
import os

import numpy as np
import pytest

from helpers import check_parity
from oracle import sto as osto
from robotoc_amd import problems as pr
from robotoc_amd.grid import (ContactSequence, Event, anymal_trot_sequence, correct_time_steps, discretize, jump_sto_sequence)
from robotoc_amd.types import BUF_DIR, BUF_KKT, BUF_STEP, GRID_IMPACT, GRID_LIFT, Grid, Records, anymal_dims

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _event_times(cs):
    return np.array([e.time for e in cs.events])


def _moved(cs, ts):
    return ContactSequence(list(cs.phase_dimf), [Event(e.kind, float(t), e.sto, e.impact_dimf) for e, t in zip(cs.events, ts)])


def _cases():
    """(name, N, T, t0, contact sequence): the jump of BASELINE configs[2], a trot with STO on every event, a late start"""
    yield "jump", 40, 0.8, 0.0, jump_sto_sequence(ground_time=0.31, flying_time=0.2)
    trot = anymal_trot_sequence(t0=0.11, swing=0.2, double_support=0.1, cycles=1)
    for e in trot.events:
        e.sto = True
    yield "trot_sto", 40, 0.8, 0.0, trot
    yield "jump_t0", 30, 0.6, 0.05, jump_sto_sequence(ground_time=0.25, flying_time=0.15)


def _random_event_times(rng, cs, t0, T, scale=0.02):
    ts = _event_times(cs) + scale * rng.uniform(-1, 1, len(cs.events))
    assert (np.diff(np.concatenate([[t0], ts, [t0 + T]])) > 0.01).all()
    return ts


def test_time_steps_restatement_matches_the_reference_discretisation():
    rng = np.random.default_rng(3)
    for name, N, T, t0, cs in _cases():
        grids = discretize(N, T, t0, cs, phase_based=True)
        for _ in range(5):
            ts = _random_event_times(rng, cs, t0, T)
            want = np.array([g.dt for g in correct_time_steps(grids, T, t0, _moved(cs, ts))])
            got = osto.correct_time_steps(grids, t0, T, ts)
            assert np.abs(got - want).max() <= 1e-15, name
            # the dwell times are the event-time differences (STOConstraints::computeDwellTimes)
            ev = osto.event_grids(grids)
            assert len(ev) == len(ts)


def test_eval_kkt_restatement_matches_the_reference_sources():
    """oracle/sto.py: eval_kkt against SwitchingTimeOptimization::evalKKT itself (its STOConstraints, an STO cost component,
    the regularisation), on the rows' initialisation (what the reference's entry point offers)."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(11)
    for name, N, T, t0, cs in _cases():
        grids = discretize(N, T, t0, cs, phase_based=True)
        n, nev = len(grids), len(cs.events)
        ts = _event_times(cs)
        t = np.concatenate([[t0], t0 + np.cumsum([g.dt for g in grids[:-1]])])
        min_dwell = rng.uniform(0.02, 0.08, nev + 1)
        h0, q0 = rng.uniform(-1, 1, n), np.abs(rng.uniform(-1, 1, n)) + 0.1
        w, tref = rng.uniform(0.5, 2.0, nev), ts + 0.03 * rng.uniform(-1, 1, nev)
        h_ref, q_ref = h0.copy(), q0.copy()
        lt_ref, qd_ref, perf = ref.sto_eval_kkt(grids, t, h_ref, q_ref, min_dwell, barrier=1.0e-2, sto_reg=0.3, cost_w=w, cost_tref=tref)
        con = osto.init_constraints(t0, T, ts, min_dwell, 1.0e-2)
        h, q = h0.copy(), q0.copy()
        lt, qd, err = osto.eval_kkt(grids, h, q, t0, T, ts, min_dwell, 1.0e-2, con, sto_reg=0.3, cost_lt=w * (ts - tref), cost_qtt=w)
        check_parity(name + " lt", float(np.abs(lt - lt_ref).max()), 1e-12)
        check_parity(name + " Qtt", float(np.abs(qd - qd_ref).max()), 1e-12)
        check_parity(name + " h", float(np.abs(h - h_ref).max()), 1e-12)
        check_parity(name + " Qtt scattered", float(np.abs(q - q_ref).max()), 1e-12)
        check_parity(name + " kkt_error", abs(err - perf[0]) / max(perf[0], 1.0), 1e-12)


def test_step_sizes_and_integration_restatement_match_the_reference_iteration():
    """The STO iteration fixture (the reference's SwitchingTimeOptimization inside OCPSolver::updateSolution, dwell-time rows
    off their initialisation): from the reference's switching-time directions, oracle/sto.py reproduces the rows' residual /
    cmpl / directions, both step sizes' STO part, the next event times and the next slacks / duals."""
    z = np.load(os.path.join(G, "ref_anymal_jump_sto_solver_iteration.npz"))
    grids = [Grid(*[int(v) for v in row], float(dt)) for row, dt in zip(z["grid"], z["grid_dt"])]
    barrier, tau, reg, t0, T = z["sto_scalars"]
    ts = z["sto_event_times"]
    con = np.zeros((6, 3))
    con[0], con[1] = z["sto_slack"], z["sto_dual"]
    n = len(grids)
    lt, qd, err = osto.eval_kkt(grids, np.zeros(n), np.zeros(n), t0, T, ts, z["sto_min_dwell"], barrier, con, sto_reg=reg)
    check_parity("lt", float(np.abs(lt - z["sto_lt_qtt"][0]).max()), 1e-13)
    check_parity("Qtt", float(np.abs(qd - z["sto_lt_qtt"][1]).max()), 1e-13)
    check_parity("rows' KKT error", abs(float(np.sum(con[2] ** 2) + np.sum(con[3] ** 2)) - z["sto_perf"][1]), 1e-13)
    dts = osto.event_dts(grids, z["sto_dts"][:, 0])
    ps, ds = osto.step_sizes(con, dts, tau)
    check_parity("rows after computeStepSizes", float(np.abs(con - z["sto_con_direction"]).max()), 1e-13)
    # the solver's steps are the minimum over the stages and the STO rows: never larger than the rows' own
    assert z["steps"][0] <= ps + 1e-15 and z["steps"][1] <= ds + 1e-15
    ts1 = osto.integrate(ts, con, dts, z["steps"][0], z["steps"][1])
    check_parity("event times", float(np.abs(ts1 - z["sto_event_times_out"]).max()), 1e-14)
    check_parity("rows after integrateSolution", float(np.abs(con - z["sto_con_out"]).max()), 1e-13)
    # the time steps the fixture's grid carries are correctTimeSteps of its event times
    check_parity("time steps", float(np.abs(osto.correct_time_steps(grids, t0, T, ts) - z["grid_dt"]).max()), 1e-15)


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["jump", "trot_sto", "jump_t0"])
def test_gpu_sto_entry_points_match_the_restatement(case):
    from robotoc_amd import capi
    name, N, T, t0, cs = next(c for c in _cases() if c[0] == case)
    grids = discretize(N, T, t0, cs, phase_based=True)
    dims = anymal_dims()
    n, nev, batch = len(grids), len(cs.events), 70   # more than one 64-thread block
    rng = np.random.default_rng(5)
    ts = np.array([_random_event_times(rng, cs, t0, T) for _ in range(batch)])
    min_dwell = rng.uniform(0.02, 0.08, nev + 1)
    barrier, tau, reg = 1.0e-3, 0.995, 0.05
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.sto_set_problem(t0, T, ts, min_dwell, barrier, tau)
        ctx.sto_set_regularization(reg)
        # correctTimeSteps: every instance its own time steps
        dt = ctx.sto_time_steps()
        want = np.array([osto.correct_time_steps(grids, t0, T, ts[b]) for b in range(batch)])
        check_parity("time steps", float(np.abs(dt - want).max()), 1e-15)
        assert np.abs(dt[0] - dt[1]).max() > 1e-4   # really per instance
        # initConstraints
        ctx.sto_init_constraints()
        con = np.array([osto.init_constraints(t0, T, ts[b], min_dwell, barrier) for b in range(batch)])
        check_parity("rows after initConstraints", float(np.abs(ctx.sto_constraint_data() - con).max()), 1e-15)
        # rows off their initialisation, STO cost terms, then evalKKT on random h / Qtt
        slack, dual = rng.uniform(0.05, 0.4, (batch, nev + 1)), rng.uniform(0.002, 0.05, (batch, nev + 1))
        ctx.sto_set_slack_dual(slack, dual)
        con[:, 0], con[:, 1] = slack, dual
        clt, cqt = rng.uniform(-1, 1, (batch, nev)), rng.uniform(0.1, 1, (batch, nev))
        ctx.sto_set_cost_terms(clt, cqt)
        kkt = pr.make_kkt_batch_tiled(L, grids, batch, unique=5)
        ctx.upload(BUF_KKT, kkt)
        K = Records(L, "kkt")
        k = kkt.copy()
        sc = K.f(k, "scal")
        lt_w, qd_w, err_w = [], [], []
        for b in range(batch):
            h, q = sc[b, :, 2].copy(), sc[b, :, 0].copy()
            lt, qd, err = osto.eval_kkt(grids, h, q, t0, T, ts[b], min_dwell, barrier, con[b], reg, clt[b], cqt[b])
            sc[b, :, 2], sc[b, :, 0] = h, q
            lt_w.append(lt), qd_w.append(qd), err_w.append(err)
        kkt_err0 = ctx.kkt_error()   # the dms part (sqrt), as rtoc_newton_iteration leaves it ahead of the STO term
        ctx.sto_eval_kkt_device()
        lt, qd, err = ctx.sto_kkt_terms()
        check_parity("lt", float(np.abs(lt - np.array(lt_w)).max()), 1e-13)
        check_parity("Qtt", float(np.abs(qd - np.array(qd_w)).max()), 1e-13)
        check_parity("squared STO KKT term", float(np.abs(err / np.array(err_w) - 1.0).max()), 1e-12)
        check_parity("h, Qtt scattered", float(np.abs(ctx.download_records(BUF_KKT, "kkt") - k).max()), 1e-13)
        check_parity("rows after evalKKT", float(np.abs(ctx.sto_constraint_data() - con).max()), 1e-14)
        # computeStepSizes from random switching-time directions in the direction records
        D = Records(L, "dir")
        d = D.zeros(batch, n)
        D.f(d, "dts")[..., 0] = 0.05 * rng.uniform(-1, 1, (batch, n))
        ctx.upload(BUF_DIR, d)
        steps0 = np.tile([0.9, 0.8], (batch, 1))
        steps0[::3] = [0.01, 0.02]   # some instances limited by the stages, not by the STO rows
        ctx.upload(BUF_STEP, steps0)
        ctx.sto_compute_step_sizes()
        steps_w = steps0.copy()
        for b in range(batch):
            ps, ds = osto.step_sizes(con[b], osto.event_dts(grids, D.f(d[b], "dts")[:, 0]), tau)
            steps_w[b] = np.minimum(steps_w[b], [ps, ds])
        steps = ctx.download(BUF_STEP, (batch, 2))
        check_parity("step sizes", float(np.abs(steps - steps_w).max()), 1e-14)
        assert (steps[:, 0] < 0.9).any() and (steps[::3, 0] == 0.01).all()
        check_parity("rows after computeStepSizes", float(np.abs(ctx.sto_constraint_data() - con).max()), 1e-13)
        # integrateSolution
        ctx.sto_integrate_solution()
        ts1 = np.array([osto.integrate(ts[b], con[b], osto.event_dts(grids, D.f(d[b], "dts")[:, 0]), steps[b, 0], steps[b, 1]) for b in range(batch)])
        check_parity("event times", float(np.abs(ctx.sto_event_times() - ts1).max()), 1e-15)
        check_parity("rows after integrateSolution", float(np.abs(ctx.sto_constraint_data() - con).max()), 1e-14)
        # and the next correctTimeSteps follows the new event times
        ctx.sto_correct_time_steps()
        want = np.array([osto.correct_time_steps(grids, t0, T, ts1[b]) for b in range(batch)])
        check_parity("time steps after the update", float(np.abs(ctx.sto_time_steps() - want).max()), 1e-15)
        assert (ctx.status() == 0).all() and np.isfinite(kkt_err0).all()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_sto_api_errors():
    from robotoc_amd import capi
    dims = anymal_dims()
    _, N, T, t0, cs = next(iter(_cases()))
    grids = discretize(N, T, t0, cs, phase_based=True)
    ctx = capi.Context(dims, len(grids), 2, 0)
    try:
        ctx.set_grid(grids)
        md = np.array([0.1, 0.1, 0.1])
        with pytest.raises(capi.RtocError):
            ctx.sto_set_problem(t0, T, np.array([0.3]), md[:2])            # one event time for a grid with two events
        with pytest.raises(capi.RtocError):
            ctx.sto_set_problem(t0, T, np.array([0.5, 0.3]), md)           # events out of order
        with pytest.raises(capi.RtocError):
            ctx.sto_set_problem(t0, T, np.array([0.3, 0.9]), md)           # beyond the horizon
        with pytest.raises(capi.RtocError):
            ctx.sto_set_problem(t0, T, np.array([0.3, 0.5]), md, barrier_param=0.0)
        ctx.sto_set_problem(t0, T, np.array([0.3, 0.5]), md)
        # a grid with other events switches the STO problem off until it is set again
        ctx.set_grid(discretize(N, T, t0, ContactSequence([12]), phase_based=False))
        with pytest.raises(capi.RtocError):
            ctx.sto_event_times()
    finally:
        ctx.close()
