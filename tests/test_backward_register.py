"""RTOC_OPT_BACKWARD_REGISTER: the register-resident backward kernel (riccati_backward_rv.hpp: one wavefront per OCP instance,
P+ / s+ in MFMA accumulators, the stage record by LDS-DMA) against the CPU oracle and against the role-split kernel, on the
GPU through the C ABI.  The horizon of the trot has every grid-point kind the kernel meets or hands over: regular, lift and
impact grid points (the kernel's own), two switching-constraint grid points (one-stage launches of the tile-split kernel, P+ / s+
handed over through the Riccati records in both directions) and the terminal record.

Tolerance: SURVEY 8c's 1e-9 per stage and field, as tests/test_gpu_parity.py (both kernels re-associate the products of
backward_riccati_recursion_factorizer.cpp:31-91 on the f64 matrix cores; observed errors are printed)."""
import numpy as np
import pytest

from helpers import compare_direction, compare_riccati
from robotoc_amd import problems as pr
from robotoc_amd.types import BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, OPT_BACKWARD_REGISTER, Records

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _sweep(ctx, kkt, dx0, register):
    ctx.set_backward_register(register)
    ctx.upload(BUF_KKT, kkt)
    ctx.upload(BUF_DX0, dx0)
    ctx.upload(BUF_RIC, np.full((kkt.shape[0], kkt.shape[1], ctx.L.ric.stride), np.nan))   # poison: every field compared must be WRITTEN
    ctx.riccati_backward()
    ctx.riccati_forward()
    return ctx.status(), ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir")


@pytest.mark.parametrize("mode", ["factory", "dynamics"])
def test_register_kernel_reproduces_the_oracle_on_the_trot(oracle, mode):
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch = 9
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode=mode)
        dx0 = pr.make_dx0(L, batch)
        st, ric, d = _sweep(ctx, kkt, dx0, True)
        assert ctx.get_option(OPT_BACKWARD_REGISTER) == 1
        R, D = Records(L, "ric"), Records(L, "dir")
        ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
        st_ref = oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        assert (st == st_ref).all(), (st, st_ref)
        worst = 0.0
        for b in range(batch):
            worst = max(worst, compare_riccati(L, grids, ric[b], ric_ref[b], TOL, "register inst %d" % b, check_sto=False))
            worst = max(worst, compare_direction(L, grids, d[b], d_ref[b], TOL, "register inst %d" % b))
        # P exactly symmetric (upper tiles computed, the rest mirrored), as the role-split kernel leaves it
        P = R.f(ric, "P")
        assert np.array_equal(P, np.swapaxes(P, -1, -2))
        # the switching-time fields of a grid without switching-time optimisation are written as zeros
        for f in ("Psi", "Phi"):
            assert not R.f(ric[:, :-1], f).any()
        print("register kernel vs oracle (%s): worst rel err %.3e" % (mode, worst))
        # ... and against the role-split kernel on the same context: same recursion, other summation order
        st2, ric2, d2 = _sweep(ctx, kkt, dx0, False)
        assert (st2 == st).all()
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric2[b], TOL, "register vs role-split inst %d" % b, check_sto=False)
        # the structured-Fxx form (RTOC_OPT_FXX_STRUCTURE: taken when every record has the state-equation structure) and the dense
        # form of the register kernel, each against the oracle; where the records are structured they are two arithmetic paths
        structured = ctx.check_fxx_structure()
        ctx.set_fxx_structure(1)
        st3, ric3, d3 = _sweep(ctx, kkt, dx0, True)
        assert (st3 == st_ref).all()
        for b in range(batch):
            compare_riccati(L, grids, ric3[b], ric_ref[b], TOL, "register dense inst %d" % b, check_sto=False)
            compare_direction(L, grids, d3[b], d_ref[b], TOL, "register dense inst %d" % b)
        assert np.array_equal(ric3, ric) != structured
        print("records structured: %s" % structured)
    finally:
        ctx.close()


def test_register_kernel_on_grids_that_start_or_end_with_a_switching_constraint(oracle):
    """Segment boundaries at both ends of the horizon: a switching constraint on the last control grid point (the terminal record is
    then written by the one-stage launch) and on grid point 0 (no register segment behind it); and a grid of impacts only."""
    from robotoc_amd import capi
    dims, grids0, _ = pr.config_anymal_trot()
    n = len(grids0)
    variants = []
    for name, where in (("constraint on the last control grid point", n - 2), ("constraint on grid point 0", 0)):
        gs = pr.config_anymal_trot()[1]
        gs[where].dims = 6
        gs[where].switching_constraint = 1
        variants.append((name, gs))
    gs = pr.config_anymal_trot()[1]
    for g in gs:   # no switching constraint anywhere: one register segment from the terminal record to grid point 0
        g.dims = 0
        g.switching_constraint = 0
    variants.append(("no switching constraint", gs))
    for name, grids in variants:
        batch = 3
        ctx = capi.Context(dims, len(grids), batch, 0)
        try:
            L = ctx.L
            ctx.set_grid(grids)
            kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
            dx0 = pr.make_dx0(L, batch)
            st, ric, d = _sweep(ctx, kkt, dx0, True)
            R, D = Records(L, "ric"), Records(L, "dir")
            ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
            st_ref = oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
            assert (st == st_ref).all(), (name, st, st_ref)
            for b in range(batch):
                compare_riccati(L, grids, ric[b], ric_ref[b], TOL, "%s inst %d" % (name, b), check_sto=False)
                compare_direction(L, grids, d[b], d_ref[b], TOL, "%s inst %d" % (name, b))
        finally:
            ctx.close()


def test_register_kernel_flags_an_indefinite_control_hessian(oracle):
    """RTOC_STAT_QUU_NOT_SPD from the register kernel where the oracle raises it (riccati_factorizer.cpp:50 is a Debug assert)."""
    from robotoc_amd import capi
    from robotoc_amd.types import STAT_QUU_NOT_SPD
    dims, grids, _ = pr.config_anymal_trot()
    batch = 4
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
        K = Records(L, "kkt")
        K.f(kkt[2, 40], "Quu")[...] = -np.eye(dims.nu)   # instance 2, a regular grid point inside the first register segment
        dx0 = pr.make_dx0(L, batch)
        st, _, _ = _sweep(ctx, kkt, dx0, True)
        assert st[2] & STAT_QUU_NOT_SPD and not (st[[0, 1, 3]] & STAT_QUU_NOT_SPD).any(), st
    finally:
        ctx.close()


def test_register_sweep_repeats_bit_for_bit():
    """No hand-offs between waves in this kernel, but DMA landing order and the deferred P stores are its own protocol: 20 sweeps
    of 4096 distinct instances, every Riccati record compared with the first run's on the device."""
    import torch
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch, n = 4096, len(grids)
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.set_backward_register(True)
        z = lambda w: torch.zeros((batch, n, getattr(L, w).stride), dtype=torch.float64, device="cuda:0")
        kkt = pr.make_kkt_batch_unique(L, grids, batch, seed=5, backend="torch", device="cuda:0", out=z("kkt"))
        dx0 = pr.make_dx0_unique(L, batch, seed=5, backend="torch", device="cuda:0").contiguous()
        ric, d = z("ric"), z("dir")
        for b_, t_ in ((BUF_KKT, kkt), (BUF_DX0, dx0), (BUF_RIC, ric), (BUF_DIR, d)):
            ctx.bind(b_, t_.data_ptr())
        torch.cuda.synchronize()
        first = None
        for rep in range(20):
            ric.fill_(float("nan"))
            torch.cuda.synchronize()
            ctx.riccati_backward()
            ctx.sync()
            assert int((ctx.status() != 0).sum()) == 0
            if first is None:
                first = ric.clone()
                nx, o = 2 * dims.nv, L.ric.off
                assert bool(torch.isfinite(first[:, :, o[0]:o[0] + nx * nx]).all())   # P, s of every grid point written
                assert bool(torch.isfinite(first[:, :, o[1]:o[1] + nx]).all())
            else:
                ne = first.view(torch.int64) != ric.view(torch.int64)
                ne &= ~(torch.isnan(first) & torch.isnan(ric))   # fields this grid never writes stay NaN in both
                assert not bool(ne.any()), "run %d: %d words differ" % (rep, int(ne.sum()))
    finally:
        ctx.close()


def test_register_kernel_flags_a_bound_record_rewritten_behind_the_runtime(oracle):
    """rtoc_bind + RTOC_OPT_FXX_STRUCTURE = 0: the device check runs once after the bind; a host that then rewrites an Fxx in place (the
    runtime cannot see it) must not get a silently wrong factorisation from the structured form -- the kernel verifies the rows it
    does not multiply and raises RTOC_STAT_FXX_UNSTRUCTURED on that instance; after rtoc_check_fxx_structure the dense form runs."""
    import torch
    from robotoc_amd import capi
    from robotoc_amd.types import STAT_FXX_UNSTRUCTURED
    dims, grids, _ = pr.config_anymal_trot()
    batch, n = 8, len(grids)
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt_h = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
        kkt = torch.from_numpy(kkt_h).to("cuda:0")
        ctx.bind(BUF_KKT, kkt.data_ptr())
        torch.cuda.synchronize()
        ctx.riccati_backward()
        assert (ctx.status() == 0).all() and ctx.check_fxx_structure()
        ctx.riccati_backward()   # (the structured form, chosen by the check above)
        assert (ctx.status() == 0).all()
        nx, o = 2 * dims.nv, L.kkt.off[0]
        # instance 5, grid point 17: a stray entry in a structured row of Fxx (row 9, column 30), written through torch
        kkt[5, 17, o + 9 + 30 * nx] = 0.125
        kkt_h[5, 17, o + 9 + 30 * nx] = 0.125
        torch.cuda.synchronize()
        ctx.clear_status()
        ctx.riccati_backward()
        st = ctx.status()
        assert st[5] & STAT_FXX_UNSTRUCTURED and not (np.delete(st, 5) & STAT_FXX_UNSTRUCTURED).any(), st
        # ... and a wrong diagonal value (a != the record's own a) is caught as well
        kkt[5, 17, o + 9 + 30 * nx] = 0.0
        kkt[2, 3, o + 10 + 10 * nx] = 0.75
        torch.cuda.synchronize()
        ctx.clear_status()
        ctx.riccati_backward()
        st = ctx.status()
        assert st[2] & STAT_FXX_UNSTRUCTURED and not (np.delete(st, 2) & STAT_FXX_UNSTRUCTURED).any(), st
        kkt[2, 3, o + 10 + 10 * nx] = 1.0
        kkt[5, 17, o + 9 + 30 * nx] = 0.125
        torch.cuda.synchronize()
        # the caller does what the status word asks for: the check refuses the structure, the dense form reproduces the oracle
        assert not ctx.check_fxx_structure()
        ctx.clear_status()
        ctx.upload(BUF_DX0, pr.make_dx0(L, batch))
        ctx.riccati_backward()
        ctx.riccati_forward()
        assert (ctx.status() == 0).all()
        ric, d = ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir")
        R, D = Records(L, "ric"), Records(L, "dir")
        ric_ref, d_ref = R.zeros(batch, n), D.zeros(batch, n)
        oracle.riccati_sweep_batch(L, grids, kkt_h.copy(), ric_ref, d_ref, dx0=pr.make_dx0(L, batch))
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric_ref[b], TOL, "rewritten record inst %d" % b, check_sto=False)
            compare_direction(L, grids, d[b], d_ref[b], TOL, "rewritten record inst %d" % b)
    finally:
        ctx.close()
