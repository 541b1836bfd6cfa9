"""RTOC_OPT_CONDENSE_REGISTER: the register-chained condensation kernel (condense_rv.hpp: one wavefront per contact grid point, the
saddle inverse read once into MFMA accumulators, the products of condenseContactDynamics chained through register layouts) against
the CPU oracle and against the role-split kernel, on the GPU through the C ABI.  The horizons hold every grid-point kind the kernel
meets or hands over: contact phases with dimf = 12 / 6 / 0, lift grid points, grid points with a switching constraint (its own), impact
grid points (condense_kernel through CondArgs::stage_list) and, on the jump, per-instance time steps.

Tolerance: 1e-9 relative per grid point and field, as tests/test_gpu_parity.py::test_sqp_iteration_hot_path (both kernels
re-associate the products of contact_dynamics.cpp:55-164 on the f64 matrix cores; observed errors are printed)."""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_KKT, OPT_CONDENSE_REGISTER, Records, joint_limit_rows

pytestmark = pytest.mark.gpu
KKT_FIELDS = ["Fxx", "Fvu", "Qxx", "Qxu", "Quu", "Fx", "lx", "lu", "hx", "hu", "fx", "scal", "Phix", "Phiu", "Phit", "Pres"]
CDD_FIELDS = ["MJtJinv", "MJtJinv_dIDCdqv", "MJtJinv_IDC", "laf", "haf", "Qxu_passive", "Quu_passive_topRight", "lu_passive", "Qaa"]


def _worst(R, got, want, fields, nst):
    from helpers import rel_err
    worst, where = 0.0, None
    for b in range(got.shape[0]):
        for i in range(nst - 1):
            for f in fields:
                e = rel_err(R.f(got[b, i], f), R.f(want[b, i], f), 1e-12)
                if not (e <= worst):
                    worst, where = e, (b, i, f)
    return worst, where


@pytest.mark.parametrize("cfg", ["anymal_trot", "anymal_jump_sto"])
@pytest.mark.parametrize("rows", [False, True, "acceleration"])
def test_register_condensation_reproduces_the_oracle(oracle, cfg, rows):
    """rows: none | the six joint-limit components | with JointAccelerationLower/UpperLimit, which act on Qaa.diagonal() / la ahead of
    everything the condensation reads (joint_acceleration_lower_limit.cpp:69-77)"""
    from helpers import check_parity
    from robotoc_amd import capi
    from robotoc_amd.types import VAR_A, anymal_dims
    dims, grids, _ = pr.config_anymal_trot() if cfg == "anymal_trot" else pr.config_anymal_jump_sto()
    if rows == "acceleration":
        dims = anymal_dims(nc_max=120)
    batch, n = 5, len(grids)
    kinds = {(g.type, g.dimf, g.dims > 0) for g in grids[:-1]}
    assert len(kinds) >= 4 and any(g.type == 1 for g in grids)   # contact phases of several dimensions, an impact, a switching constraint
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        assert ctx.get_option(OPT_CONDENSE_REGISTER) == 1   # the default, and no cone rows here: the register kernel runs
        kkt, cdd = pr.make_precondense_batch(L, grids, batch)
        con = pr.make_constraint_batch(L, grids, batch)
        box = joint_limit_rows(dims, acceleration=(rows == "acceleration"))
        if rows == "acceleration":   # rows that matter next to Qaa ~ O(1): dual / slack of the same order (tests/test_acceleration_limits.py)
            N0 = Records(L, "con")
            a_rows = [r for r, w in enumerate(box) if w.var == VAR_A]
            N0.f(con, "dual")[..., a_rows] *= 300.0
            N0.f(con, "cmpl")[..., a_rows] = N0.f(con, "slack")[..., a_rows] * N0.f(con, "dual")[..., a_rows] - 1.0e-3
        if rows:
            ctx.set_constraint_rows(box)
        out = {}
        for name, on in (("register", True), ("role-split", False)):
            ctx.set_condense_register(on)
            ctx.upload(BUF_KKT, kkt)
            ctx.upload(BUF_CDD, cdd)
            ctx.upload(BUF_CON, con)
            ctx.condense()
            assert (ctx.status() == 0).all()
            out[name] = (ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd"), ctx.download_records(BUF_CON, "con"))
        kk, cc, nn = kkt.copy(), cdd.copy(), con.copy()
        if rows:
            oracle.pdipm_condense_batch(L, grids, box, kk, nn, cdd=cc)
        assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
        K, Cd, Nn = Records(L, "kkt"), Records(L, "cdd"), Records(L, "con")
        for name in ("register", "role-split"):
            ek, wk = _worst(K, out[name][0], kk, KKT_FIELDS, n)
            ec, wc = _worst(Cd, out[name][1], cc, CDD_FIELDS, n)
            print("%s %s rows=%s vs oracle: condensed KKT %.2e %s, contact-dynamics data %.2e %s" % (cfg, name, rows, ek, wk, ec, wc))
            check_parity("%s condensed KKT" % name, ek, 1e-9)
            check_parity("%s contact-dynamics data" % name, ec, 1e-9)
            if rows:
                from helpers import rel_err
                check_parity("%s joint-limit rows cond" % name, rel_err(Nn.f(out[name][2], "cond"), Nn.f(nn, "cond")), 1e-11)
        # another kernel did run: the two association orders differ in the last bits somewhere
        assert not np.array_equal(out["register"][0], out["role-split"][0])
    finally:
        ctx.close()


def test_register_condensation_with_cone_rows_agrees_with_the_role_split_kernel():
    """Friction-cone rows condensed INSIDE condense_rv_kernel (the tiles of their Gram product go into the seeds and operands of the
    condensation; Qqf, Qff, lf and the rows' condensing coefficients back to the records) -- the same records as the one-kernel
    role-split condensation leaves, which condenses them in its second wave; 72 joint-limit rows beside them."""
    from helpers import check_parity, rel_err
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch, n = 6, len(grids)
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.set_constraint_rows(joint_limit_rows(dims))
        ctx.set_friction_cones(4, 3)
        kkt, cdd = pr.make_precondense_batch_unique(L, grids, batch)
        con = pr.make_constraint_batch_unique(L, grids, batch)
        cone = pr.make_cone_batch_unique(L, grids, batch, 4)
        out = {}
        for name, opt in (("register", None), ("role-split", False)):   # (the default: friction-cone rows are condensed inside the kernel)
            if opt is not None:
                ctx.set_condense_register(opt)
            for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, con), (BUF_CONE, cone)):
                ctx.upload(buf, arr)
            ctx.condense()
            assert (ctx.status() == 0).all()
            out[name] = (ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd"), ctx.download_records(BUF_CON, "con"))
        K, Cd, Nn = Records(L, "kkt"), Records(L, "cdd"), Records(L, "con")
        ek, wk = _worst(K, out["register"][0], out["role-split"][0], KKT_FIELDS, n)
        ec, wc = _worst(Cd, out["register"][1], out["role-split"][1], CDD_FIELDS + ["Qff", "Qqf", "lf"], n)
        en = max(rel_err(Nn.f(out["register"][2], f), Nn.f(out["role-split"][2], f)) for f in ("cond", "slack", "dual"))
        print("register (cone rows inside) vs one-kernel condensation: KKT %.2e %s, contact-dynamics data %.2e %s, rows %.2e" % (ek, wk, ec, wc, en))
        check_parity("condensed KKT", ek, 1e-10)
        check_parity("contact-dynamics data", ec, 1e-10)
        check_parity("constraint rows", en, 1e-12)
        assert not np.array_equal(out["register"][0], out["role-split"][0])
    finally:
        ctx.close()


def test_register_condensation_repeats_bit_for_bit():
    """One wave per work item and no atomics: 1024 instances condensed twice leave identical records."""
    from robotoc_amd import capi
    dims, grids, _ = pr.config_anymal_trot()
    batch, n = 1024, len(grids)
    ctx = capi.Context(dims, n, batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.set_constraint_rows(joint_limit_rows(dims))
        kkt, cdd = pr.make_precondense_batch_unique(L, grids, 64)
        con = pr.make_constraint_batch_unique(L, grids, 64)
        tile = lambda a: np.ascontiguousarray(np.tile(a, (batch // a.shape[0],) + (1,) * (a.ndim - 1)))
        runs = []
        for _ in range(2):
            for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_CON, con)):
                ctx.upload(buf, tile(arr))
            ctx.condense()
            assert (ctx.status() == 0).all()
            runs.append((ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd")))
        assert ctx.get_option(OPT_CONDENSE_REGISTER) == 1
        assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
        # ... and every copy of an instance got the same records
        k = runs[0][0].reshape(batch // 64, 64, n, -1)
        assert np.array_equal(k[0], k[-1])
    finally:
        ctx.close()
