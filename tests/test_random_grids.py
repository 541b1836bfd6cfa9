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
from robotoc_amd.types import (BUF_CDD, BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, GRID_IMPACT, GRID_LIFT, GRID_TERMINAL,
                               Records, anymal_dims, icub_dims)


def random_case(seed, dims=None, nmax=22):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(10, nmax))
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
    return (dims if dims is not None else anymal_dims()), discretize(N, T, 0.0, cs, phase_based=any_sto), any_sto


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
        tol = 1e-9   # SURVEY 8c (observed over all seeds: 2.2e-11 with STO grids, 9.4e-12 without)
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric_ref[b], tol, "seed %d inst %d" % (seed, b))
            compare_direction(L, grids, d[b], d_ref[b], tol, "seed %d inst %d" % (seed, b))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nv,seed", [(32, 3), (32, 8), (35, 5), (35, 11)])
def test_random_event_structures_icub(oracle, nv, seed):
    """The tile-split kernel at iCub size (the only variant for nu > 16) on random event structures."""
    from robotoc_amd import capi
    dims, grids, any_sto = random_case(seed, icub_dims(nv), nmax=14)
    batch = 2
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics", first_instance=seed)
        dx0 = pr.make_dx0(L, batch, first_instance=seed)
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_DX0, dx0)
        ctx.riccati_sweep()
        ric_ref = Records(L, "ric").zeros(batch, len(grids))
        d_ref = Records(L, "dir").zeros(batch, len(grids))
        st_ref = oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        assert (ctx.status() == st_ref).all()
        ric, d = ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir")
        tol = 1e-9   # SURVEY 8c (observed over all seeds: 2.2e-11 with STO grids, 9.4e-12 without)
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric_ref[b], tol, "nv %d seed %d inst %d" % (nv, seed, b))
            compare_direction(L, grids, d[b], d_ref[b], tol, "nv %d seed %d inst %d" % (nv, seed, b))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_event_structures_condense_expand(oracle, seed):
    """Contact-dynamics condensation / expansion on the same random structures: every mix of nf in
    {0, 6, 12}, impact stages and switching-constraint stages (condenseContactDynamics /
    condenseImpactDynamics / expand*) against the oracle."""
    from robotoc_amd import capi
    dims, grids, _ = random_case(seed)
    batch = 2
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt, cdd = pr.make_precondense_batch(L, grids, batch, first_instance=seed)
        dx0 = pr.make_dx0(L, batch, first_instance=seed)
        for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_DX0, dx0)):
            ctx.upload(buf, arr)
        # odd seeds: RTOC_OPT_CONDENSE_KEEP_QAF (Qafqv / Qafu_full stored and compared); even seeds: the default, where
        # the condensation does not store them and the expansion must still come out right
        keep = bool(seed & 1)
        ctx.set_condense_keep_qaf(keep)
        ctx.condense()
        kkt_gpu = ctx.download_records(BUF_KKT, "kkt")
        cdd_gpu = ctx.download_records(BUF_CDD, "cdd")
        ctx.riccati_sweep()
        ctx.expand(0.995)
        d_gpu = ctx.download_records(BUF_DIR, "dir")
        assert (ctx.status() == 0).all()
        kk, cc = kkt.copy(), cdd.copy()
        assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
        K, C, D = Records(L, "kkt"), Records(L, "cdd"), Records(L, "dir")
        from helpers import rel_err
        for f in ("Qxx", "Qxu", "Quu", "lx", "lu", "Fxx", "Fvu", "Fx", "Phix", "Phiu", "Pres"):
            assert rel_err(K.f(kkt_gpu, f), K.f(kk, f)) < 1e-9, (seed, f)
        for f in ("MJtJinv", "MJtJinv_dIDCdqv", "MJtJinv_IDC", "laf") + (("Qafqv", "Qafu_full") if keep else ()):
            assert rel_err(C.f(cdd_gpu, f), C.f(cc, f)) < 1e-9, (seed, f)
        ric_ref, d_ref = Records(L, "ric").zeros(batch, len(grids)), D.zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kk, ric_ref, d_ref, dx0=dx0)
        oracle.expand_batch(L, grids, cc, d_ref)
        for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu", "dnu_passive"):
            from helpers import check_parity
            check_parity("sqp directions " + f, rel_err(D.f(d_gpu, f), D.f(d_ref, f)), 1e-9)
    finally:
        ctx.close()
