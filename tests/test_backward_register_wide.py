"""RTOC_OPT_BACKWARD_REGISTER on the iCub-size shapes: the register-wide backward kernel (riccati_backward_rw.hpp: one wavefront per
OCP instance with the whole register file of its SIMD, P+ in 16 / 25 MFMA accumulator tiles, the dense rows of the structured Fxx
by LDS-DMA) against the CPU oracle and against the tile-split kernel, on the GPU through the C ABI.  The jump's horizon has every
grid-point kind the kernel meets or hands over: regular, lift and impact grid points (its own), the switching-constraint grid
point (a one-stage launch of the tile-split kernel, P+ / s+ through the Riccati records both ways) and the terminal record.

Tolerance: SURVEY 8c's 1e-9 per stage and field, as tests/test_gpu_parity.py (the kernels re-associate the products of
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


@pytest.mark.parametrize("mode", ["dynamics", "factory"])
@pytest.mark.parametrize("nv", [32, 35])
def test_register_wide_kernel_reproduces_the_oracle_on_the_icub_jump(oracle, nv, mode):
    from robotoc_amd import capi
    dims, grids, _ = pr.config_icub_jump(nv=nv)
    batch = 6
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode=mode)
        dx0 = pr.make_dx0(L, batch)
        st, ric, d = _sweep(ctx, kkt, dx0, 2)   # 2: the register-wide kernel whatever the batch size
        assert ctx.check_fxx_structure()
        assert ctx.get_option(OPT_BACKWARD_REGISTER) == 2
        R, D = Records(L, "ric"), Records(L, "dir")
        ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
        st_ref = oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        assert (st == st_ref).all(), (st, st_ref)
        worst = 0.0
        for b in range(batch):
            worst = max(worst, compare_riccati(L, grids, ric[b], ric_ref[b], TOL, "register-wide inst %d" % b, check_sto=False))
            worst = max(worst, compare_direction(L, grids, d[b], d_ref[b], TOL, "register-wide inst %d" % b))
        # P exactly symmetric on the kernel's own grid points (upper tiles computed, the rest mirrored)
        P = R.f(ric, "P")
        own = [i for i, g in enumerate(grids) if not (g.type != 1 and g.dims > 0)]
        assert np.array_equal(P[:, own], np.swapaxes(P[:, own], -1, -2))
        # the switching-time fields of a grid without switching-time optimisation are written as zeros
        for f in ("Psi", "Phi"):
            assert not R.f(ric[:, :-1], f).any()
        print("register-wide kernel vs oracle (nv %d, %s): worst rel err %.3e" % (nv, mode, worst))
        # ... and against the tile-split kernel on the same context: same recursion, another summation order (so not bit-identical:
        # that the two differ at all is the evidence that the new kernel ran)
        st2, ric2, d2 = _sweep(ctx, kkt, dx0, 0)
        assert (st2 == st).all()
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric2[b], TOL, "register-wide vs tile-split inst %d" % b, check_sto=False)
        assert not np.array_equal(ric, ric2)
        # with 1 (the default) a batch below the device's CU count keeps the tile-split kernel: bit-identical to the run with 0
        st3, ric3, _ = _sweep(ctx, kkt, dx0, 1)
        assert np.array_equal(ric3, ric2, equal_nan=True)
    finally:
        ctx.close()


@pytest.mark.parametrize("nv", [32, 35])
@pytest.mark.parametrize("horizon", ["one_stage", "two_stages", "uniform_63"])
def test_register_wide_kernel_on_ragged_horizons(oracle, nv, horizon):
    """The shortest horizons the kernel accepts (one and two stages ahead of the terminal record: its prologue / stage-end hand-over
    with nothing in between) and the longest (64 grid points: the size of its grid-kind table) -- against the oracle, batch 3,
    RTOC_OPT_BACKWARD_REGISTER = 2."""
    from robotoc_amd import capi
    from robotoc_amd.grid import uniform_grid
    from robotoc_amd.types import icub_dims
    dims = icub_dims(nv)
    grids = uniform_grid({"one_stage": 1, "two_stages": 2, "uniform_63": 63}[horizon], 0.02, dimf=12)
    batch = 3
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
        dx0 = pr.make_dx0(L, batch)
        st, ric, d = _sweep(ctx, kkt, dx0, 2)
        R, D = Records(L, "ric"), Records(L, "dir")
        ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
        st_ref = oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        assert (st == st_ref).all(), (st, st_ref)
        worst = 0.0
        for b in range(batch):
            worst = max(worst, compare_riccati(L, grids, ric[b], ric_ref[b], TOL, "%s inst %d" % (horizon, b), check_sto=False))
            worst = max(worst, compare_direction(L, grids, d[b], d_ref[b], TOL, "%s inst %d" % (horizon, b)))
        if len(grids) > 2:   # the register-wide kernel ran (needs two stages): not bit-identical to the tile-split kernel
            _, ric2, _ = _sweep(ctx, kkt, dx0, 0)
            assert not np.array_equal(ric, ric2)
        print("register-wide kernel, %s, nv %d: worst rel err %.3e" % (horizon, nv, worst))
    finally:
        ctx.close()


@pytest.mark.parametrize("nv", [32, 35])
def test_register_wide_sweep_as_a_replayed_graph_and_in_chunks(nv):
    """rtoc_riccati_sweep with the register-wide kernels (a) captured into a HIP graph and replayed (the structure check of Fxx runs
    ahead of the capture), (b) pipelined over three chunks of instances (launches with a first-instance offset): the directions and
    the Riccati records equal the plain sweep's bit for bit."""
    from robotoc_amd import capi
    dims, grids, _ = pr.config_icub_jump(nv=nv)
    batch = 7
    kkt, dx0 = None, None
    out = {}
    for name in ("plain", "graph", "chunks"):
        ctx = capi.Context(dims, len(grids), batch, 0)
        try:
            L = ctx.L
            ctx.set_grid(grids)
            ctx.set_backward_register(2)
            if kkt is None:
                kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
                dx0 = pr.make_dx0(L, batch)
            if name == "graph":
                ctx.set_graph(True)
            if name == "chunks":
                ctx.set_sweep_chunks(3)
            ctx.upload(BUF_KKT, kkt)
            ctx.upload(BUF_DX0, dx0)
            for _ in range(4 if name == "graph" else 1):   # warm-up, capture, replays
                ctx.riccati_sweep()
            assert (ctx.status() == 0).all()
            if name == "graph":
                assert ctx.graph_replay_count() >= 2
            out[name] = (ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir"))
        finally:
            ctx.close()
    for name in ("graph", "chunks"):
        assert np.array_equal(out[name][0], out["plain"][0], equal_nan=True), name
        assert np.array_equal(out[name][1], out["plain"][1], equal_nan=True), name


def test_register_wide_kernel_is_not_chosen_for_an_unstructured_fxx(oracle):
    """One stray entry in the structured half of one Fxx: the device check refuses, the tile-split kernel runs (dense Fxx), the
    oracle is reproduced."""
    from robotoc_amd import capi
    dims, grids, _ = pr.config_icub_jump(nv=32)
    batch = 3
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
        Records(L, "kkt").f(kkt[1, 7], "Fxx")[10, 40] = 0.25
        dx0 = pr.make_dx0(L, batch)
        st, ric, d = _sweep(ctx, kkt, dx0, 2)
        assert not ctx.check_fxx_structure()
        R, D = Records(L, "ric"), Records(L, "dir")
        ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
        st_ref = oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        assert (st == st_ref).all()
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric_ref[b], TOL, "unstructured inst %d" % b, check_sto=False)
            compare_direction(L, grids, d[b], d_ref[b], TOL, "unstructured inst %d" % b)
        st2, ric2, _ = _sweep(ctx, kkt, dx0, 0)
        assert np.array_equal(ric, ric2, equal_nan=True)
    finally:
        ctx.close()


def test_register_wide_kernel_flags_an_indefinite_control_hessian(oracle):
    from robotoc_amd import capi
    from robotoc_amd.types import STAT_QUU_NOT_SPD
    dims, grids, _ = pr.config_icub_jump(nv=32)
    batch = 4
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
        Records(L, "kkt").f(kkt[2, 28], "Quu")[...] = -np.eye(dims.nu)
        dx0 = pr.make_dx0(L, batch)
        st, _, _ = _sweep(ctx, kkt, dx0, 2)
        assert st[2] & STAT_QUU_NOT_SPD and not (st[[0, 1, 3]] & STAT_QUU_NOT_SPD).any(), st
    finally:
        ctx.close()


@pytest.mark.parametrize("nv", [32, 35])
def test_register_wide_sweep_repeats_bit_for_bit(nv):
    """DMA landing order, the deferred P stores and the hand-over through the Riccati records at the switching-constraint grid point:
    10 sweeps of 1024 distinct instances (the default dispatch: more instances than CUs), every Riccati record compared with the
    first run's on the device."""
    import torch
    from robotoc_amd import capi
    dims, grids, _ = pr.config_icub_jump(nv=nv)
    batch, n = 1024, len(grids)
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        z = lambda w: torch.zeros((batch, n, getattr(L, w).stride), dtype=torch.float64, device="cuda:0")   # noqa: E731
        kkt = pr.make_kkt_batch_unique(L, grids, batch, seed=5, backend="torch", device="cuda:0", out=z("kkt"))
        dx0 = pr.make_dx0_unique(L, batch, seed=5, backend="torch", device="cuda:0").contiguous()
        ric, d = z("ric"), z("dir")
        for b_, t_ in ((BUF_KKT, kkt), (BUF_DX0, dx0), (BUF_RIC, ric), (BUF_DIR, d)):
            ctx.bind(b_, t_.data_ptr())
        torch.cuda.synchronize()
        first = None
        for rep in range(10):
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
        # against the tile-split kernel on the same records (1e-9 per field): the whole batch on the device
        ctx.set_backward_register(0)
        ric2 = first.clone()
        ric.fill_(float("nan"))
        torch.cuda.synchronize()
        ctx.riccati_backward()
        ctx.sync()
        nx, o = 2 * dims.nv, L.ric.off
        for lo_, n_ in ((o[0], nx * nx), (o[1], nx)):
            a_, b_ = ric2[:, :, lo_:lo_ + n_], ric[:, :, lo_:lo_ + n_]
            scale = b_.abs().amax(dim=2, keepdim=True).clamp_min(1e-300)
            assert float(((a_ - b_).abs() / scale).max()) < 1e-8
    finally:
        ctx.close()


def test_register_wide_kernel_rechecks_a_bound_buffer_before_every_recursion(oracle):
    """The register-wide kernel never loads the structured rows of Fxx, so it cannot verify them: on a BOUND buffer
    (RTOC_OPT_FXX_STRUCTURE = 0) the device check runs again before every backward recursion.  A record rewritten in place behind
    the runtime's back switches the context to the tile-split kernel by itself -- the result is the dense kernels', bit for bit."""
    import torch
    from robotoc_amd import capi
    dims, grids, _ = pr.config_icub_jump(nv=32)
    batch, n = 5, len(grids)
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.set_backward_register(2)
        kkt_h = pr.make_kkt_batch(L, grids, batch, mode="dynamics")
        kkt = torch.from_numpy(kkt_h).to("cuda:0")
        ric = torch.zeros((batch, n, L.ric.stride), dtype=torch.float64, device="cuda:0")
        ctx.bind(BUF_KKT, kkt.data_ptr())
        ctx.bind(BUF_RIC, ric.data_ptr())
        torch.cuda.synchronize()
        ctx.riccati_backward()
        ctx.sync()
        assert (ctx.status() == 0).all()
        wide = ric.clone()
        nx, o = 2 * dims.nv, L.kkt.off[0]
        kkt[3, 12, o + 20 + 5 * nx] = -0.5   # a stray entry in a structured row
        torch.cuda.synchronize()
        ctx.riccati_backward()
        ctx.sync()
        assert (ctx.status() == 0).all()
        auto = ric.clone()
        ctx.set_backward_register(0)
        ctx.riccati_backward()
        ctx.sync()
        assert torch.equal(auto.view(torch.int64), ric.view(torch.int64))   # the tile-split kernel ran, unasked
        assert not torch.equal(auto[3].view(torch.int64), wide[3].view(torch.int64))
    finally:
        ctx.close()
