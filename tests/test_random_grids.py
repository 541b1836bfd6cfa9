"""Randomised event structures: contact sequences with random lift / impact times, contact dimensions
and STO flags go through TimeDiscretization (robotoc_amd/grid.py) and the resulting grids -- every
combination of Impact / Lift / switching-constraint / sto / sto_next stage the dispatch of
riccati_recursion.cpp:41-70 can produce -- are swept on the GPU and compared with the oracle.
CPU part: the generator only yields grids the library accepts (no GPU needed to check the validator)."""
import numpy as np
import pytest

from helpers import compare_direction, compare_riccati
from robotoc_amd import problems as pr
from robotoc_amd.grid import ContactSequence, Event, discretize
from robotoc_amd.types import (BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, GRID_IMPACT, GRID_LIFT, GRID_TERMINAL,
                               Records, anymal_dims)


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(10, 22))
    dt = 0.02
    T = N * dt
    nev = int(rng.integers(1, 5))
    # event times at least 2.3 dt apart and away from both ends
    gaps = rng.uniform(2.3, 4.0, nev + 1)
    times = np.cumsum(gaps[:-1]) * dt + rng.uniform(0.2, 0.8) * dt
    sto_on = bool(rng.integers(0, 2))
    dimf = 12
    phase_dimf = [dimf]
    events = []
    for tm in times:
        if tm + 2.5 * dt > T:
            break
        if dimf > 0 and (dimf == 12 or rng.integers(0, 2)):
            new = int(rng.choice([d for d in (0, 6) if d < dimf]))
            events.append(Event("lift", float(tm), sto=sto_on and bool(rng.integers(0, 2))))
        else:
            new = int(rng.choice([d for d in (6, 12) if d > dimf]))
            events.append(Event("impact", float(tm), sto=sto_on and bool(rng.integers(0, 2)),
                                impact_dimf=new - dimf))
        dimf = new
        phase_dimf.append(dimf)
    cs = ContactSequence(phase_dimf, events)
    any_sto = any(e.sto for e in events)
    return anymal_dims(), discretize(N, T, 0.0, cs, phase_based=any_sto), any_sto


def _valid(grids):
    n = len(grids)
    for i, g in enumerate(grids):
        if g.type == GRID_IMPACT and (i == 0 or i >= n - 2):
            return False
        if g.type == GRID_LIFT and i == 0:
            return False
        if g.dt < 0 or g.dimf > 12 or g.dims > 12:
            return False
    return grids[-1].type == GRID_TERMINAL


def test_generator_produces_varied_valid_grids():
    kinds = set()
    for seed in range(24):
        _, grids, any_sto = random_case(seed)
        assert _valid(grids), seed
        for g in grids:
            kinds.add((g.type, bool(g.sto), bool(g.sto_next), g.dims > 0))
    # impacts, lifts, switching constraints and STO stages all occur
    assert any(k[0] == GRID_IMPACT for k in kinds) and any(k[0] == GRID_LIFT for k in kinds)
    assert any(k[3] for k in kinds) and any(k[1] for k in kinds) and any(k[2] for k in kinds)
    assert len(kinds) >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("waves", [0, 1, 2, 3])
@pytest.mark.parametrize("seed", range(24))
def test_random_event_structures_match_oracle(oracle, seed, waves):
    """waves: 0 = the default kernel (role-split, 4 instances per workgroup), 1/2/3 = the one-wave,
    role-split pair and tile-split variants (rtoc.h RTOC_OPT_BACKWARD_WAVES)."""
    from robotoc_amd import capi
    dims, grids, any_sto = random_case(seed)
    batch = 2 + seed % 4    # also odd batches / partial 4-instance workgroups
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        if waves:
            ctx.set_backward_waves(waves)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics", first_instance=seed)
        dx0 = pr.make_dx0(L, batch, first_instance=seed)
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_DX0, dx0)
        ctx.riccati_sweep()
        st = ctx.status()
        ric_ref = Records(L, "ric").zeros(batch, len(grids))
        d_ref = Records(L, "dir").zeros(batch, len(grids))
        st_ref = oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        assert (st == st_ref).all(), (st, st_ref)
        ric, d = ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir")
        tol = 1e-7 if any_sto else 1e-8
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric_ref[b], tol, "seed %d inst %d" % (seed, b))
            compare_direction(L, grids, d[b], d_ref[b], tol, "seed %d inst %d" % (seed, b))
    finally:
        ctx.close()
