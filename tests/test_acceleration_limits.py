"""JointAccelerationLowerLimit / JointAccelerationUpperLimit in the hot path (reference
src/constraints/joint_acceleration_lower_limit.cpp:69-95, joint_acceleration_upper_limit.cpp:69-95): RTOC_VAR_A rows of
rtoc_set_constraint_rows.  Unlike the six q / v / u components their condensation does NOT commute with the contact-dynamics
condensation -- they add to Qaa.diagonal() and la, which condenseContactDynamics reads (contact_dynamics.cpp:68-86) -- so the
kernel eliminates them between MJtJinv_dIDCdqv and Qafqv, and the expansion reads da.  GPU vs the C oracle (which
tests/test_constraints_vs_reference.py pins to the reference's own sources), on both condensation pipelines, together with the
six other joint-limit components and the friction cones."""
import numpy as np
import pytest

from helpers import check_parity, compare_direction, rel_err
from robotoc_amd import problems as pr
from robotoc_amd.types import (BUF_CDD, BUF_CON, BUF_CONE, BUF_DIR, BUF_DX0, BUF_KKT, BUF_STEP, VAR_A, Records, anymal_dims,
                               joint_limit_rows)

MC, CD = 4, 3


def _problem(batch):
    _, grids, _ = pr.config_anymal_trot()
    dims = anymal_dims(nc_max=120)   # 8 x 12 joint-limit rows + 20 cone rows, padded
    rows = joint_limit_rows(dims, acceleration=True)
    assert sum(1 for r in rows if r.var == VAR_A) == 24
    return dims, grids, rows


def _data(L, grids, rows, batch):
    kkt, cdd = pr.make_precondense_batch(L, grids, batch)
    con = pr.make_constraint_batch(L, grids, batch)
    N = Records(L, "con")
    # acceleration rows that matter next to Qaa ~ O(1): dual / slack of the same order
    a_rows = [r for r, w in enumerate(rows) if w.var == VAR_A]
    N.f(con, "dual")[..., a_rows] *= 300.0
    N.f(con, "cmpl")[..., a_rows] = N.f(con, "slack")[..., a_rows] * N.f(con, "dual")[..., a_rows] - 1.0e-3
    return kkt, cdd, con, pr.make_cone_batch(L, grids, batch, MC), pr.make_dx0(L, batch)


def test_oracle_acceleration_rows_change_the_condensed_hessians(oracle):
    """the rows reach the condensed Quu / Qxx through Qaa (and lu / lx through la): the oracle-side sanity of the test data"""
    dims, grids, rows = _problem(2)
    L = oracle.layout(dims)
    kkt, cdd, con, cone, dx0 = _data(L, grids, rows, 2)
    out = []
    for use in (rows, [w for w in rows if w.var != VAR_A]):
        kk, cc, nn = kkt.copy(), cdd.copy(), con.copy()
        oracle.pdipm_condense_batch(L, grids, use, kk, nn, cc)
        assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
        out.append(kk)
    K = Records(L, "kkt")
    assert rel_err(K.f(out[0], "Quu"), K.f(out[1], "Quu")) > 1e-3
    assert rel_err(K.f(out[0], "lu"), K.f(out[1], "lu")) > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("split", [1, 0])
def test_gpu_hot_path_with_acceleration_limits(oracle, split):
    from robotoc_amd import capi
    batch = 3
    dims, grids, rows = _problem(batch)
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.set_condense_split(split)
        kkt, cdd, con, cone, dx0 = _data(L, grids, rows, batch)
        ctx.set_constraint_rows(rows)
        ctx.set_friction_cones(MC, CD)
        for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, con), (BUF_CONE, cone), (BUF_DX0, dx0)):
            ctx.upload(buf, arr)
        ctx.condense()
        kkt_gpu, cdd_gpu = ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd")
        ctx.riccati_backward()
        ctx.riccati_forward()
        ctx.expand(0.995)
        steps_gpu = ctx.download(BUF_STEP, (batch, 2))
        con_exp = ctx.download_records(BUF_CON, "con")
        d_gpu = ctx.download_records(BUF_DIR, "dir")
        ctx.update()
        con_upd = ctx.download_records(BUF_CON, "con")
        assert (ctx.status() == 0).all()
        # oracle: condenseSlackAndDual (joint limits incl. acceleration, cones) -> contact dynamics -> Riccati -> expansions
        kk, cc, nn = kkt.copy(), cdd.copy(), con.copy()
        oracle.pdipm_condense_batch(L, grids, rows, kk, nn, cc)
        oracle.cone_condense_batch(L, grids, MC, CD, cone, kk, cc, nn)
        K, Cd, R, D, N = (Records(L, w) for w in ("kkt", "cdd", "ric", "dir", "con"))
        check_parity("Qaa diagonal after the acceleration rows", rel_err(Cd.f(cdd_gpu, "Qaa"), Cd.f(cc, "Qaa")), 1e-13)
        assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
        for f in ("Qxx", "Qxu", "Quu", "lx", "lu", "Fxx", "Fvu", "Fx"):
            check_parity("condensed " + f, rel_err(K.f(kkt_gpu, f), K.f(kk, f)), 1e-9)
        for f in ("MJtJinv", "MJtJinv_dIDCdqv", "MJtJinv_IDC", "laf"):
            check_parity("contact dynamics data " + f, rel_err(Cd.f(cdd_gpu, f), Cd.f(cc, f)), 1e-9)
        ric_ref, d_ref = R.zeros(batch, len(grids)), D.zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kk, ric_ref, d_ref, dx0=dx0)
        oracle.expand_batch(L, grids, cc, d_ref)
        for b in range(batch):
            compare_direction(L, grids, d_gpu[b], d_ref[b], 1e-8, "inst %d" % b)
        for f in ("daf", "dbetamu"):
            check_parity("expansion " + f, rel_err(D.f(d_gpu, f), D.f(d_ref, f)), 1e-8)
        steps_ref = oracle.pdipm_expand_batch(L, grids, rows, nn, d_ref, 0.995)
        oracle.cone_expand_batch(L, grids, MC, CD, cone, nn, d_ref, 0.995, steps_ref)
        a_rows = [r for r, w in enumerate(rows) if w.var == VAR_A]
        for f in ("cond", "dslack", "ddual"):
            check_parity("pdipm " + f, rel_err(N.f(con_exp, f), N.f(nn, f)), 1e-7)
            check_parity("acceleration rows " + f, rel_err(N.f(con_exp, f)[..., a_rows], N.f(nn, f)[..., a_rows]), 1e-7)
        check_parity("fraction-to-boundary steps", float(np.abs(steps_gpu / steps_ref - 1.0).max()), 1e-7)
        oracle.pdipm_update_batch(L, grids, rows, nn, steps_gpu)
        oracle.cone_update_batch(L, grids, MC, CD, nn, steps_gpu)
        for f in ("slack", "dual"):
            check_parity("update " + f, rel_err(N.f(con_upd, f), N.f(nn, f)), 1e-9)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_acceleration_rows_are_refused_without_contacts():
    """rtoc_set_constraint_rows: RTOC_VAR_A rows belong to the contact path (level 0, shapes with contacts)"""
    from robotoc_amd import capi
    from robotoc_amd.types import BoxRow, Dims
    ctx = capi.Context(Dims(7, 7, 0, 0, 0, 16), 4, 1, 0)
    try:
        with pytest.raises(capi.RtocError):
            ctx.set_constraint_rows([BoxRow(VAR_A, 0, -1, 0)])
    finally:
        ctx.close()
    dims, grids, rows = _problem(1)
    ctx = capi.Context(dims, len(grids), 1, 0)
    try:
        with pytest.raises(capi.RtocError):
            ctx.set_constraint_rows([BoxRow(VAR_A, 6, -1, 1)])   # not an acceleration-level row
        ctx.set_constraint_rows(rows)
    finally:
        ctx.close()
