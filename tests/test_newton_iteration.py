"""rtoc_newton_iteration: one Newton / SQP iteration of the batch as a single launch sequence
with the convergence test on the device (OCPSolver::updateSolution + the KKTError() < kkt_tol test of
OCPSolver::solve, src/solver/ocp_solver.cpp:111-145, 200-206).  GPU only:
  * against the ORACLE-side sequence of the same iteration (KKT error -> PDIPM / cone condensation -> contact
    dynamics condensation -> Riccati sweep -> expansions -> step sizes -> convergence mask -> slack / dual update
    -> SplitSolution::integrate), every output buffer compared;
  * against the same steps issued one by one through the C ABI, bit for bit;
instances whose KKT error is below the tolerance keep their iterate and are counted."""
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
    o = L.sol.off[0]
    sol[..., o + 3:o + 7] /= np.linalg.norm(sol[..., o + 3:o + 7], axis=-1, keepdims=True)  # q on the manifold
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
        err = ref.kkt_error()  # OCPSolver::KKTError() itself (the sqrt is taken on the device)
        assert err.max() > 1.2 * err.min()
        tol = float(np.sort(err)[batch // 2 - 1]) * (1 + 1e-12)  # the smaller half counts as converged
        nconv = int((err < tol).sum())
        assert 0 < nconv < batch
        ref.condense()
        ref.riccati_sweep()
        ref.expand(tau)
        steps = ref.download(BUF_STEP, (batch, 2))
        assert (steps > 0).all()
        steps[err < tol] = 0.0
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
        conv = err < tol
        S = Records(fused.L, "sol")
        for f in ("q", "v", "a", "f", "lmd", "gmm", "beta", "mu"):  # (u is zeroed on impact grids whatever the step)
            assert np.array_equal(S.f(sol1[conv], f), S.f(sol0[conv], f)), f  # converged instances keep their iterate
        assert not np.array_equal(S.f(sol1[~conv], "v"), S.f(sol0[~conv], "v"))
    finally:
        ref.close()
        fused.close()


@pytest.mark.gpu
def test_newton_iteration_against_the_oracle_sequence(oracle):
    """The fused iteration vs the CPU oracle running OCPSolver::updateSolution's hot path step by step
    (ocp_solver.cpp:118-142 downstream of the linearisation) on the same pre-condensation records."""
    from helpers import rel_err
    batch, tau = 6, 0.995
    fused, sol0 = _context(batch)
    try:
        L = fused.L
        dims, grids, _ = pr.config_anymal_trot()
        rows = joint_limit_rows(dims)
        kkt, cdd = (fused.download_records(b, w) for b, w in ((BUF_KKT, "kkt"), (BUF_CDD, "cdd")))
        con = fused.download_records(BUF_CON, "con")
        cone = pr.make_cone_batch(L, grids, batch, MC)
        dx0 = pr.make_dx0(L, batch)
        # oracle: KKTError() on the freshly linearised records
        err_ref = oracle.kkt_error(L, grids, kkt, cdd, con, rows, MC, CD, 5)
        err_gpu = fused.kkt_error()
        assert np.allclose(err_gpu, err_ref, rtol=1e-12), (err_gpu, err_ref)
        tol = float(np.sort(err_ref)[batch // 2 - 1]) * (1 + 1e-9)
        conv = err_ref < tol
        assert 0 < conv.sum() < batch
        fused.newton_iteration(tol, tau)
        assert fused.converged_count() == int(conv.sum())
        assert (fused.status() == 0).all()
        # oracle: condenseSlackAndDual (box rows, cones) -> contact dynamics -> sweep -> expansions
        kk, cc, nn = kkt.copy(), cdd.copy(), con.copy()
        oracle.pdipm_condense_batch(L, grids, rows, kk, nn)
        oracle.cone_condense_batch(L, grids, MC, CD, cone, kk, cc, nn)
        assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
        R, D, N, S = (Records(L, w) for w in ("ric", "dir", "con", "sol"))
        ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kk, ric_ref, d_ref, dx0=dx0)
        oracle.expand_batch(L, grids, cc, d_ref)
        steps_ref = oracle.pdipm_expand_batch(L, grids, rows, nn, d_ref, tau)
        oracle.cone_expand_batch(L, grids, MC, CD, cone, nn, d_ref, tau, steps_ref)
        steps_ref[conv] = 0.0
        steps_gpu = fused.download(BUF_STEP, (batch, 2))
        from helpers import check_parity
        live = steps_ref > 0
        check_parity("step sizes", float(np.abs(steps_gpu[live] / steps_ref[live] - 1.0).max()) if live.any() else 0.0, 1e-9)
        assert (steps_gpu[~live] == 0.0).all()
        d_gpu = fused.download_records(BUF_DIR, "dir")
        worst = 0.0
        for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu", "dnu_passive"):
            e = rel_err(D.f(d_gpu, f), D.f(d_ref, f))
            worst = max(worst, e)
            check_parity("direction " + f, e, 1e-9)
        # the updates with the GPU's own step sizes (ratios of direction entries: compared above at 1e-9)
        oracle.pdipm_update_batch(L, grids, rows, nn, steps_gpu)
        oracle.cone_update_batch(L, grids, MC, CD, nn, steps_gpu)
        con_gpu = fused.download_records(BUF_CON, "con")
        for f in ("slack", "dual", "dslack", "ddual", "cond"):
            e = rel_err(N.f(con_gpu, f), N.f(nn, f))
            worst = max(worst, e)
            check_parity("rows " + f, e, 1e-9)
        sol_ref = sol0.copy()
        oracle.integrate_solution_batch(L, grids, steps_gpu, d_ref, sol_ref)
        sol_gpu = fused.download_records(BUF_SOL, "sol")
        for f in ("q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu", "nu_passive", "xi"):
            e = rel_err(S.f(sol_gpu, f), S.f(sol_ref, f))
            worst = max(worst, e)
            check_parity("solution " + f, e, 1e-9)
        for f in ("q", "v", "a", "f", "lmd", "gmm", "beta", "mu"):
            assert np.array_equal(S.f(sol_gpu[conv], f), S.f(sol0[conv], f)), f
        print("newton iteration vs oracle sequence: worst rel err %.3e" % worst)
    finally:
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
    to the scan's tolerance (1e-8 relative, step sizes included)."""
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
            from helpers import check_parity
            check_parity("scan vs serial " + f, float(np.linalg.norm(x - y) / np.linalg.norm(x)), 1e-8)
        sa, sb = serial.download(BUF_STEP, (batch, 2)), scan.download(BUF_STEP, (batch, 2))
        check_parity("scan vs serial step sizes", float(np.abs(sa / sb - 1.0).max()), 1e-8)
        for buf, which in ((BUF_CON, "con"), (BUF_SOL, "sol")):
            x, y = serial.download_records(buf, which), scan.download_records(buf, which)
            check_parity("scan vs serial " + which, float(np.linalg.norm(x - y) / np.linalg.norm(x)), 1e-8)
    finally:
        serial.close()
        scan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scan", [False, True])
def test_graph_replay_is_bit_identical(scan):
    """RTOC_OPT_GRAPH: the Newton iteration and the sweep replayed from captured hipGraphs give the bits of the plain
    launch sequences -- across the warm-up call, the capturing call and the replays, after a re-upload, and after a
    change of configuration (new tolerance arguments, an option) that forces a re-capture."""
    batch, tau = 2, 0.995
    plain, _ = _context(batch)
    graph, _ = _context(batch)
    try:
        graph.set_graph(True)
        plain.set_backward_scan(scan)
        graph.set_backward_scan(scan)
        L = plain.L
        dims, grids, _ = pr.config_anymal_trot()
        kkt, cdd = pr.make_precondense_batch(L, grids, batch)
        con = pr.make_constraint_batch(L, grids, batch)
        sol = np.random.default_rng(5).uniform(-1, 1, (batch, len(grids), L.sol.stride))

        def restore(c):
            for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, con), (BUF_SOL, sol)):
                c.upload(buf, arr)
        for it, tol in enumerate((0.0, 0.0, 0.0, 0.0, 1e30, 1e30, 1e30)):
            restore(plain)
            restore(graph)
            plain.newton_iteration(tol, tau)
            graph.newton_iteration(tol, tau)
            for buf, which in ((BUF_DIR, "dir"), (BUF_CON, "con"), (BUF_SOL, "sol")):
                assert np.array_equal(graph.download_records(buf, which), plain.download_records(buf, which)), (it, which)
            assert graph.converged_count() == plain.converged_count() == (batch if tol > 1 else 0)
        # the records were re-uploaded before every iteration (the documented loop): the graph must have been REPLAYED all the
        # same -- call 0 warms up, call 1 captures, 2 and 3 replay; the new tolerance re-captures at 4, then 5 and 6 replay
        n_newton = graph.graph_replay_count()
        assert n_newton >= 6, n_newton
        # the sweep alone, on the condensed records of the last iteration
        for it in range(4):
            plain.riccati_sweep()
            graph.riccati_sweep()
            assert np.array_equal(graph.download_records(BUF_DIR, "dir"), plain.download_records(BUF_DIR, "dir"))
        assert graph.graph_replay_count() - n_newton >= 3   # warm-up, capture (+ launch), two replays
        assert plain.graph_replay_count() == 0
        assert (graph.status() == 0).all()
    finally:
        plain.close()
        graph.close()
