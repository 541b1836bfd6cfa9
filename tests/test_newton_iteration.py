"""rtoc_newton_iteration: one Newton / SQP iteration of the batch as a single launch sequence
with the convergence test on the device.  GPU only: the fused entry point against the same steps issued one by
one through the C ABI (each of which has its own parity test against the oracle), bit for bit; instances whose
KKT error is below the tolerance keep their iterate and are counted."""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import (BUF_CDD, BUF_CON, BUF_CONE, BUF_DIR, BUF_DX0, BUF_KKT, BUF_SOL, BUF_STEP, Records,
                               joint_limit_rows)

MC, CD = 4, 3


def _context(batch):
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    ctx = capi.Context(dims, len(grids), batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    ctx.set_constraint_rows(joint_limit_rows(dims))
    ctx.set_friction_cones(MC, CD)
    kkt, cdd = pr.make_precondense_batch(L, grids, batch)
    # instances with visibly different KKT errors: scale the residual-like fields of every other instance
    K = Records(L, "kkt")
    for b in range(0, batch, 2):
        for f in ("Fx", "lx", "lu"):
            K.f(kkt[b], f)[...] *= 1e-3
        Records(L, "cdd").f(cdd[b], "IDC")[...] *= 1e-3
    rng = np.random.default_rng(5)
    sol = rng.uniform(-1, 1, (batch, len(grids), L.sol.stride))
    for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, pr.make_constraint_batch(L, grids, batch)),
                     (BUF_CONE, pr.make_cone_batch(L, grids, batch, MC)), (BUF_DX0, pr.make_dx0(L, batch)),
                     (BUF_SOL, sol)):
        ctx.upload(buf, arr)
    return ctx, sol


@pytest.mark.gpu
def test_newton_iteration_equals_the_sequence_of_its_steps():
    batch, tau = 6, 0.995
    ref, sol0 = _context(batch)
    fused, _ = _context(batch)
    try:
        err = np.sqrt(ref.kkt_error())
        assert err.max() > 1.2 * err.min()
        tol = float(np.sort(err)[batch // 2 - 1]) * (1 + 1e-12)  # the smaller half counts as converged
        nconv = int((err <= tol).sum())
        assert 0 < nconv < batch
        ref.condense()
        ref.riccati_sweep()
        ref.expand(tau)
        steps = ref.download(BUF_STEP, (batch, 2))
        assert (steps > 0).all()
        steps[err <= tol] = 0.0
        ref.upload(BUF_STEP, steps)
        ref.update()
        ref.integrate_solution()
        fused.newton_iteration(tol, tau)
        assert fused.converged_count() == nconv
        assert (fused.status() == 0).all() and (ref.status() == 0).all()
        for buf, which in ((BUF_DIR, "dir"), (BUF_CON, "con"), (BUF_SOL, "sol")):
            assert np.array_equal(fused.download_records(buf, which), ref.download_records(buf, which)), which
        assert np.array_equal(fused.download(BUF_STEP, (batch, 2)), steps)
        sol1 = fused.download_records(BUF_SOL, "sol")
        conv = err <= tol
        S = Records(fused.L, "sol")
        for f in ("q", "v", "a", "f", "lmd", "gmm", "beta", "mu"):  # (u is zeroed on impact grids whatever the step)
            assert np.array_equal(S.f(sol1[conv], f), S.f(sol0[conv], f)), f  # converged instances keep their iterate
        assert not np.array_equal(S.f(sol1[~conv], "v"), S.f(sol0[~conv], "v"))
    finally:
        ref.close()
        fused.close()


@pytest.mark.gpu
def test_newton_iteration_argument_checks():
    from robotoc_amd import capi
    ctx, _ = _context(2)
    try:
        with pytest.raises(capi.RtocError):
            ctx.newton_iteration(-1.0)
        with pytest.raises(capi.RtocError):
            ctx.newton_iteration(1e-6, tau=1.5)
        ctx.newton_iteration(1e30)          # everything "converged"
        assert ctx.converged_count() == 2
    finally:
        ctx.close()


@pytest.mark.gpu
def test_newton_iteration_with_the_horizon_scan():
    """The whole SQP hot path with RTOC_OPT_BACKWARD_SCAN (condensation -> both recursions as scans -> expansion ->
    step sizes -> updates): directions, step sizes and the updated iterate agree with the serial recursions
    to the scan's tolerance (1e-8 relative; 1e-6 on the step sizes, which are ratios of direction entries)."""
    batch, tau = 3, 0.995
    serial, _ = _context(batch)
    scan, _ = _context(batch)
    try:
        scan.set_backward_scan(True)
        serial.newton_iteration(0.0, tau)
        scan.newton_iteration(0.0, tau)
        assert (serial.status() == 0).all() and (scan.status() == 0).all()
        L = serial.L
        D = Records(L, "dir")
        a, b = serial.download_records(BUF_DIR, "dir"), scan.download_records(BUF_DIR, "dir")
        assert not np.array_equal(a, b)  # really a different arithmetic path
        for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu"):
            x, y = D.f(a, f), D.f(b, f)
            assert np.linalg.norm(x - y) <= 1e-8 * np.linalg.norm(x), f
        sa, sb = serial.download(BUF_STEP, (batch, 2)), scan.download(BUF_STEP, (batch, 2))
        assert np.allclose(sa, sb, rtol=1e-6, atol=0)
        for buf, which in ((BUF_CON, "con"), (BUF_SOL, "sol")):
            x, y = serial.download_records(buf, which), scan.download_records(buf, which)
            assert np.linalg.norm(x - y) <= 1e-7 * np.linalg.norm(x), which
    finally:
        serial.close()
        scan.close()
