"""Pins two pieces of caller-side logic to the REFERENCE'S OWN SOURCES, compiled where they
lie into oracle/_ref/librtoc_ref_td.so (oracle/Makefile.ref):
  * robotoc_amd/grid.py: discretize() -- the grid tables every test and the bench hand to rtoc_set_grid, and what the mesh
    refinement of OCPSolver::solve (ocp_solver.cpp:184-199) re-runs -- against robotoc's TimeDiscretization::discretize /
    correctTimeSteps / maxTimeStep (src/ocp/time_discretization.cpp), over a stand-in ContactSequence that only holds
    event times and STO flags;
  * the line-search filter: the plain-Python restatement used to write tests/golden/ref_line_search_filter.npz against
    robotoc's LineSearchFilter (src/line_search/line_search_filter.cpp).  The GPU test replays that fixture through
    rtoc_line_search_filter -- neither the reference nor any oracle in that loop.
Runs wherever the library exists (built here; the prebuilt .so travels with the snapshot)."""
import os

import numpy as np
import pytest

from robotoc_amd import capi
from robotoc_amd import grid as G
from robotoc_amd.types import anymal_dims
from test_random_grids import random_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = pytest.importorskip("oracle.ref")
needs_ref = pytest.mark.skipif(not (os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librtoc_ref_td.so")) or os.path.isdir(ref.REFERENCE)),
                               reason="oracle/_ref not built and /root/reference absent")


def _check(N, T, t, cs, phase_based):
    ev = [(e.kind, e.time, e.sto) for e in cs.events]
    tab, dt, _, max_dt = ref.discretize(T, N, t, ev, phase_based)
    g = G.discretize(N, T, t, cs, phase_based=phase_based)
    assert len(g) == tab.shape[0]
    for i, (gi, row) in enumerate(zip(g, tab)):
        assert (gi.type, gi.sto, gi.sto_next, gi.switching_constraint) == (row[0], row[2], row[3], row[4]), (i, row)
        if i < len(g) - 1:
            assert gi.num_grids_in_phase == row[6], (i, row)
        assert abs(gi.dt - dt[i]) <= 1e-15
    assert abs(G.max_time_step(g) - max_dt) <= 1e-15
    return g


@needs_ref
def test_named_configurations_match_the_reference_discretization():
    _check(40, 0.8, 0.0, G.anymal_trot_sequence(), False)
    _check(40, 0.8, 0.0, G.jump_sto_sequence(), True)
    _check(8, 0.16, 0.0, G.anymal_trot_sequence(t0=0.03, swing=0.05, double_support=0.03, cycles=1), False)


@needs_ref
def test_random_event_sequences_match_the_reference_discretization():
    rng = np.random.default_rng(7)
    n = 0
    for seed in range(60):
        N = int(rng.integers(10, 40))
        dt = 0.02
        T = N * dt
        t = float(rng.uniform(0.0, 0.05))
        times = t + np.cumsum(rng.uniform(2.3, 6.0, 4)) * dt
        events, dimf, phase_dimf = [], 12, [12]
        for tm in times:
            if tm + 2.5 * dt > t + T:
                break
            if dimf > 0 and (dimf == 12 or rng.integers(0, 2)):
                new = int(rng.choice([d for d in (0, 6) if d < dimf]))
                events.append(G.Event("lift", float(tm), sto=bool(rng.integers(0, 2))))
            else:
                new = int(rng.choice([d for d in (6, 12) if d > dimf]))
                events.append(G.Event("impact", float(tm), sto=bool(rng.integers(0, 2)), impact_dimf=new - dimf))
            dimf = new
            phase_dimf.append(dimf)
        cs = G.ContactSequence(phase_dimf, events)
        for pb in (False, True):
            _check(N, T, t, cs, pb)
            n += 1
    assert n == 120


@needs_ref
def test_mesh_refinement_rule():
    """ocp_solver.cpp:184-199: re-discretise when the largest time step exceeds max_dt_mesh; after the switching times
    moved, the phase-based grid has unequal steps and the re-discretisation restores dt <= T / N per interval."""
    cs = G.jump_sto_sequence(ground_time=0.31, flying_time=0.2)
    g0 = G.discretize(40, 0.8, 0.0, cs, phase_based=True)
    # the optimiser moved the lift-off 80 ms later: the first phase stretches
    cs.events[0].time += 0.08
    cs.events[1].time += 0.08
    g1 = G.correct_time_steps(g0, 0.8, 0.0, cs)
    assert G.max_time_step(g1) > 0.8 / 40 + 1e-6 and len(g1) == len(g0)
    g2, refined = G.mesh_refinement(g1, 40, 0.8, 0.0, cs, max_dt_mesh=0.8 / 40 + 1e-6)
    assert refined and G.max_time_step(g2) <= G.max_time_step(g1)
    ev = [(e.kind, e.time, e.sto) for e in cs.events]
    tab, dt, _, _ = ref.discretize(0.8, 40, 0.0, ev, True)
    assert len(g2) == tab.shape[0] and np.allclose([x.dt for x in g2], dt, rtol=0, atol=1e-15)
    g3, refined = G.mesh_refinement(g0, 40, 0.8, 0.0, G.jump_sto_sequence(), max_dt_mesh=0.05)
    assert not refined and g3 is g0


# ---------------------------------------------------------------------------------------------------------------------
def py_filter_try(filt, cost, viol, cr, vr):
    """LineSearchFilter::isAccepted + augment (line_search_filter.cpp:26-60) on a Python list of pairs"""
    ok = not filt or any((cost < c - cr * v) or (viol < (1.0 - vr) * v) for c, v in filt)
    if ok:
        filt[:] = [(c, v) for c, v in filt if not (c <= cost and v <= viol)] + [(cost, viol)]
    return int(ok)


def filter_sequences(seed=11, batch=64, steps=40):
    rng = np.random.default_rng(seed)
    base = rng.uniform(1.0, 10.0, batch)
    cost = base[None, :] * (1.0 + 0.3 * rng.standard_normal((steps, batch))) * np.linspace(1.0, 0.3, steps)[:, None]
    viol = np.abs(rng.standard_normal((steps, batch))) * np.linspace(1.0, 0.05, steps)[:, None]
    mask = (rng.uniform(size=(steps, batch)) < 0.85).astype(np.int32)
    return cost, viol, mask


@needs_ref
def test_filter_restatement_and_fixture_match_the_reference_filter():
    cost, viol, mask = filter_sequences()
    steps, batch = cost.shape
    acc = np.zeros((steps, batch), dtype=np.int32)
    for b in range(batch):
        f, pf = ref.LineSearchFilter(0.005, 0.005), []
        for s in range(steps):
            if mask[s, b]:
                acc[s, b] = f.try_step(cost[s, b], viol[s, b])
                assert py_filter_try(pf, cost[s, b], viol[s, b], 0.005, 0.005) == acc[s, b]
    assert 0.2 < acc.mean() < 0.9
    path = os.path.join(ROOT, "tests", "golden", "ref_line_search_filter.npz")
    if os.environ.get("RTOC_WRITE_GOLDEN"):
        np.savez_compressed(path, cost=cost, violation=viol, mask=mask, accepted=acc)
    fx = np.load(path)
    assert np.array_equal(fx["accepted"], acc) and np.array_equal(fx["cost"], cost)


@pytest.mark.gpu
def test_gpu_line_search_filter_replays_the_reference_fixture():
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_line_search_filter.npz"))
    cost, viol, mask, acc = fx["cost"], fx["violation"], fx["mask"], fx["accepted"]
    steps, batch = cost.shape
    ctx = capi.Context(anymal_dims(), 4, batch, 0)
    for rep in range(2):  # second round after clearHistory: same decisions again
        ctx.line_search_clear()
        for s in range(steps):
            got = ctx.line_search_filter(cost[s], viol[s], mask[s])
            assert np.array_equal(got, acc[s]), (rep, s)
    # capacity: a strictly improving-in-one-coordinate sequence never erases anything; the newest entries are kept
    ctx.line_search_clear()
    one = np.ones(batch)
    for k in range(capi.LINE_SEARCH_FILTER_CAPACITY + 8):
        assert ctx.line_search_filter((100.0 - k) * one, (1.0 + k) * one).all()
    ctx.close()
