"""PDIPM rows of the linearised friction cones (reference
src/constraints/friction_cone.cpp:194-268): condensation into Qqq, Qqf, Qff, lq, lf BEFORE the
contact-dynamics condensation, expansion of dslack / ddual with the fraction-to-boundary step sizes,
slack / dual update.  CPU: the oracle against the closed form in numpy.  GPU: through the C ABI
against the oracle, stand-alone and inside the full SQP hot path (joint limits + cones)."""
import numpy as np
import pytest

from helpers import rel_err
from robotoc_amd import problems as pr
from robotoc_amd.types import (BUF_CDD, BUF_CON, BUF_CONE, BUF_DIR, BUF_DX0, BUF_KKT, BUF_STEP, GRID_TERMINAL,
                               Records, cone_dgdf_off, joint_limit_rows)

MC, CD = 4, 3  # ANYmal: 4 point contacts


def _blocks(L, cone_rec, k):
    nv = L.dims.nv
    off = cone_dgdf_off(nv, MC)
    dq = cone_rec[k * 5 * nv:(k + 1) * 5 * nv].reshape(nv, 5).T
    df = cone_rec[off + k * 15:off + (k + 1) * 15].reshape(3, 5).T
    return dq, df


def test_oracle_cone_condense_closed_form(oracle):
    dims, grids, _ = pr.config_anymal_trot()
    L = oracle.layout(dims)
    batch = 2
    kkt, cdd = pr.make_precondense_batch(L, grids, batch)
    con = pr.make_constraint_batch(L, grids, batch)
    cone = pr.make_cone_batch(L, grids, batch, MC)
    K, C, N = Records(L, "kkt"), Records(L, "cdd"), Records(L, "con")
    nv, row0 = dims.nv, dims.nc_max - 5 * MC
    kkt_ref, cdd_ref, con_ref = kkt.copy(), cdd.copy(), con.copy()
    touched = 0
    for b in range(batch):
        for i, g in enumerate(grids):
            if g.type == GRID_TERMINAL:
                continue
            for k in range(g.dimf // CD):
                dq, df = _blocks(L, cone[b, i], k)
                r = slice(row0 + 5 * k, row0 + 5 * k + 5)
                slack, dual = N.f(con_ref[b, i], "slack")[r], N.f(con_ref[b, i], "dual")[r]
                cond = (dual * N.f(con_ref[b, i], "residual")[r] - N.f(con_ref[b, i], "cmpl")[r]) / slack
                N.f(con_ref[b, i], "cond")[r] = cond
                W = np.diag(dual / slack)
                K.f(kkt_ref[b, i], "lx")[:nv] += dq.T @ cond
                C.f(cdd_ref[b, i], "lf")[k * CD:k * CD + 3] += df.T @ cond
                K.f(kkt_ref[b, i], "Qxx")[:nv, :nv] += dq.T @ W @ dq
                C.f(cdd_ref[b, i], "Qqf")[:, k * CD:k * CD + 3] += dq.T @ W @ df
                C.f(cdd_ref[b, i], "Qff")[k * CD:k * CD + 3, k * CD:k * CD + 3] += df.T @ W @ df
                touched += 1
    assert touched > 100
    oracle.cone_condense_batch(L, grids, MC, CD, cone, kkt, cdd, con)
    assert np.allclose(kkt, kkt_ref, rtol=1e-13, atol=1e-13)
    assert np.allclose(cdd, cdd_ref, rtol=1e-13, atol=1e-13)
    assert np.allclose(con, con_ref, rtol=1e-13, atol=1e-13)


def test_oracle_cone_expand_closed_form(oracle):
    dims, grids, _ = pr.config_anymal_trot()
    L = oracle.layout(dims)
    batch = 2
    con = pr.make_constraint_batch(L, grids, batch)
    cone = pr.make_cone_batch(L, grids, batch, MC)
    D, N = Records(L, "dir"), Records(L, "con")
    rng = np.random.default_rng(9)
    d = D.zeros(batch, len(grids))
    d[...] = 0.2 * rng.uniform(-1, 1, d.shape)
    nv, row0, tau = dims.nv, dims.nc_max - 5 * MC, 0.995
    con_ref = con.copy()
    steps_ref = np.ones((batch, 2))
    for b in range(batch):
        for i, g in enumerate(grids):
            if g.type == GRID_TERMINAL:
                continue
            for k in range(g.dimf // CD):
                dq, df = _blocks(L, cone[b, i], k)
                r = slice(row0 + 5 * k, row0 + 5 * k + 5)
                slack, dual = N.f(con_ref[b, i], "slack")[r], N.f(con_ref[b, i], "dual")[r]
                dslack = -dq @ D.f(d[b, i], "dx")[:nv] - df @ D.f(d[b, i], "daf")[nv + k * CD:nv + k * CD + 3] \
                    - N.f(con_ref[b, i], "residual")[r]
                ddual = -(dual * dslack + N.f(con_ref[b, i], "cmpl")[r]) / slack
                N.f(con_ref[b, i], "dslack")[r] = dslack
                N.f(con_ref[b, i], "ddual")[r] = ddual
                for v, dv, col in ((slack, dslack, 0), (dual, ddual, 1)):
                    f = -tau * v / dv
                    f = f[(f > 0) & (f < 1)]
                    if f.size:
                        steps_ref[b, col] = min(steps_ref[b, col], f.min())
    steps = np.ones((batch, 2))
    oracle.cone_expand_batch(L, grids, MC, CD, cone, con, d, tau, steps)
    assert np.allclose(con, con_ref, rtol=1e-13, atol=1e-13)
    assert np.allclose(steps, steps_ref, rtol=1e-14)
    assert (steps < 1).any()


@pytest.mark.gpu
def test_gpu_sqp_hot_path_with_cones_and_joint_limits(oracle):
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch = 3
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt, cdd = pr.make_precondense_batch(L, grids, batch)
        con = pr.make_constraint_batch(L, grids, batch)
        cone = pr.make_cone_batch(L, grids, batch, MC)
        dx0 = pr.make_dx0(L, batch)
        rows = joint_limit_rows(dims)
        ctx.set_constraint_rows(rows)
        ctx.set_friction_cones(MC, CD)
        for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, con), (BUF_CONE, cone), (BUF_DX0, dx0)):
            ctx.upload(buf, arr)
        ctx.condense()
        kkt_gpu = ctx.download_records(BUF_KKT, "kkt")
        cdd_gpu = ctx.download_records(BUF_CDD, "cdd")
        ctx.riccati_backward()
        ctx.riccati_forward()
        ctx.expand(0.995)
        steps_gpu = ctx.download(BUF_STEP, (batch, 2))
        con_exp = ctx.download_records(BUF_CON, "con")
        d_gpu = ctx.download_records(BUF_DIR, "dir")
        ctx.update()
        con_upd = ctx.download_records(BUF_CON, "con")
        assert (ctx.status() == 0).all()
        # oracle: Constraints::condenseSlackAndDual (joint limits, cones) -> contact dynamics -> Riccati
        kk, cc, nn = kkt.copy(), cdd.copy(), con.copy()
        oracle.pdipm_condense_batch(L, grids, rows, kk, nn)
        oracle.cone_condense_batch(L, grids, MC, CD, cone, kk, cc, nn)
        # the cone terms alone (before the dynamics condensation consumes them): Qff, Qqf, lf inputs
        C = Records(L, "cdd")
        for f in ("Qff", "Qqf", "lf"):
            assert rel_err(C.f(cdd_gpu, f), C.f(cc, f)) < 1e-12, f
        assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
        K = Records(L, "kkt")
        for f in ("Qxx", "Qxu", "Quu", "lx", "lu", "Fxx", "Fvu", "Fx"):
            assert rel_err(K.f(kkt_gpu, f), K.f(kk, f)) < 1e-9, f
        R, D, N = Records(L, "ric"), Records(L, "dir"), Records(L, "con")
        ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kk, ric_ref, d_ref, dx0=dx0)
        oracle.expand_batch(L, grids, cc, d_ref)
        for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu"):
            assert rel_err(D.f(d_gpu, f), D.f(d_ref, f)) < 1e-7, f
        steps_ref = oracle.pdipm_expand_batch(L, grids, rows, nn, d_ref, 0.995)
        oracle.cone_expand_batch(L, grids, MC, CD, cone, nn, d_ref, 0.995, steps_ref)
        for f in ("cond", "dslack", "ddual"):
            assert rel_err(N.f(con_exp, f), N.f(nn, f)) < 1e-7, f
        assert np.allclose(steps_gpu, steps_ref, rtol=1e-6), (steps_gpu, steps_ref)
        oracle.pdipm_update_batch(L, grids, rows, nn, steps_gpu)
        oracle.cone_update_batch(L, grids, MC, CD, nn, steps_gpu)
        for f in ("slack", "dual"):
            assert rel_err(N.f(con_upd, f), N.f(nn, f)) < 1e-9, f
    finally:
        ctx.close()


@pytest.mark.gpu
def test_friction_cone_rows_need_room_in_the_constraint_record():
    from robotoc_amd import capi
    from robotoc_amd.types import anymal_dims
    dims = anymal_dims(nc_max=72)  # exactly the 72 joint-limit rows: no room for cone rows
    _, grids, _ = pr.config_anymal_trot()
    ctx = capi.Context(dims, len(grids), 1, 0)
    try:
        ctx.set_grid(grids)
        ctx.set_constraint_rows(joint_limit_rows(dims))
        with pytest.raises(capi.RtocError):
            ctx.set_friction_cones(MC, CD)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cones", ["friction", "none"])
def test_split_and_fused_condensation_agree(cones):
    """RTOC_OPT_CONDENSE_SPLIT: MJtJinv (+ the cone rows) in their own kernel vs everything in one kernel."""
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch = 3
    out = {}
    for split in (1, 0):
        ctx = capi.Context(dims, len(grids), batch, 0)
        try:
            L = ctx.L
            ctx.set_grid(grids)
            ctx.set_condense_split(split)
            kkt, cdd = pr.make_precondense_batch(L, grids, batch)
            con = pr.make_constraint_batch(L, grids, batch)
            ctx.set_constraint_rows(joint_limit_rows(dims))
            for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, con)):
                ctx.upload(buf, arr)
            if cones == "friction":
                ctx.set_friction_cones(MC, CD)
                ctx.upload(BUF_CONE, pr.make_cone_batch(L, grids, batch, MC))
            ctx.condense()
            assert (ctx.status() == 0).all()
            out[split] = (ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd"),
                          ctx.download_records(BUF_CON, "con"))
        finally:
            ctx.close()
    for a, b in zip(out[1], out[0]):
        assert np.allclose(a, b, rtol=1e-11, atol=1e-12)
