"""The iterate-independent entries of evalKKT, which rtoc_contact_eval_kkt writes with the setZero of the KKT records
(init_records_kernel, robotoc_amd/csrc/contact_eval_kkt.hpp): the diagonals dt W of Qqq (joints), Qvv and Quu of a
ConfigurationSpaceCost (reference src/cost/configuration_space_cost.cpp:274-324 stage, :381-470 impact, :343-378 terminal) and
Fqq = I, Fqv = dt I of linearizeStateEquation (src/dynamics/state_equation.cpp:29-40; impact_state_equation.cpp: dt = 0; none on
the terminal grid).  Checked entry by entry on an OCP without inequality rows, where nothing else touches these blocks: iCub with
nv = 35 (nu = 29 is ODD: 16-byte stores straddle the columns of Quu), iCub nv = 32 and ANYmal, over intermediate, lift, impact and
terminal grid points."""
import numpy as np
import pytest

from robotoc_amd import robot_model as rm
from robotoc_amd.types import BUF_KKT, BUF_SOL, GRID_IMPACT, Dims, Records


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["icub", "icub32", "anymal"])
def test_constant_entries_of_eval_kkt(robot):
    from robotoc_amd import capi
    from robotoc_amd.grid import ContactSequence, Event, contact_masks, discretize
    m = rm.load_named(robot)
    nv, nq, nu, nx = m.nv, m.nq, m.nu, 2 * m.nv
    dimf = m.max_dimf
    dims = Dims(nv, nu, nv - nu, dimf, dimf, 0)
    full = (1 << m.ncontacts) - 1
    grids = discretize(12, 0.36, 0.0, ContactSequence([dimf, 0, dimf], [Event("lift", 0.11), Event("impact", 0.23, impact_dimf=dimf)]))
    n, batch = len(grids), 3
    assert any(g.type == GRID_IMPACT for g in grids)
    ctx = capi.Context(dims, n, batch, 0)
    try:
        ctx.set_grid(grids)
        ctx.set_robot_model(m)
        masks = contact_masks(grids, [full, 0, full], [full])
        rot = np.tile(np.eye(3)[None, None], (n, m.ncontacts, 1, 1))
        ctx.set_contact_schedule(masks, np.zeros((n, m.ncontacts, 3)), rot)
        rng = np.random.default_rng(3)
        w = {k: rng.uniform(0.5, 2.0, sz) for k, sz in (("wq", nv), ("wv", nv), ("wa", nv), ("wu", nu), ("wqT", nv), ("wvT", nv),
                                                         ("wqI", nv), ("wvI", nv), ("wdvI", nv))}
        q_ref = rm.random_configuration(m, rng, 0.3)[0]
        ctx.set_configuration_cost(q_ref, np.zeros(nv), np.zeros(nu), w["wq"], w["wv"], w["wa"], w["wu"], w["wqT"], w["wvT"],
                                   q_weight_impact=w["wqI"], v_weight_impact=w["wvI"], dv_weight_impact=w["wdvI"])
        x0 = np.tile(np.concatenate([rm.random_configuration(m, rng, 0.3)[0], np.zeros(nv)]), (batch, 1))
        ctx.set_initial_state(x0)
        S = Records(ctx.L, "sol")
        sol = S.zeros(batch, n)
        for b in range(batch):
            for i in range(n):
                q, v, a = rm.random_configuration(m, rng, 0.3)
                S.f(sol[b, i], "q")[:nq], S.f(sol[b, i], "v")[:], S.f(sol[b, i], "a")[:] = q, v, a
        ctx.upload(BUF_SOL, sol)
        ctx.upload(BUF_KKT, np.full(ctx.shape("kkt"), np.nan))   # every entry of the record is written, whatever was there
        ctx.contact_eval_kkt()
        ctx.sync()
        kkt = ctx.download_records(BUF_KKT, "kkt")
        assert (ctx.status() == 0).all() and np.isfinite(kkt).all()
        K = Records(ctx.L, "kkt")
        nb = 6 if m.floating_base else 0
        for i, g in enumerate(grids):
            terminal, impact = i == n - 1, g.type == GRID_IMPACT
            scale = 1.0 if (terminal or impact) else g.dt
            Wq = w["wqT"] if terminal else w["wqI"] if impact else w["wq"]
            Wv = w["wvT"] if terminal else w["wvI"] if impact else w["wv"]
            for b in range(batch):
                Qxx, Quu, Fxx = K.f(kkt[b, i], "Qxx"), K.f(kkt[b, i], "Quu"), K.f(kkt[b, i], "Fxx")
                want = np.diag(np.concatenate([scale * Wq, scale * Wv]))
                got = Qxx.copy()
                got[:nb, :nb] = want[:nb, :nb] = 0.0      # the base block: J^T W J of the SE(3) difference (contact_cost_kernel)
                assert np.array_equal(got, want), (robot, i, b)
                assert np.array_equal(Quu, np.diag(g.dt * w["wu"]) if not (terminal or impact) else np.zeros((nu, nu))), (robot, i, b)
                top = np.zeros((nv, nx))
                if not terminal:
                    top[:, :nv] = np.eye(nv)
                    top[:, nv:] = (0.0 if impact else g.dt) * np.eye(nv)
                gotF = Fxx[:nv].copy()
                gotF[:nb, :nb] = top[:nb, :nb] = 0.0      # Fqq of the base: Jlog6 (state_equation_lin_kernel)
                assert np.array_equal(gotF, top), (robot, i, b)
                assert not Fxx[nv:].any(), (robot, i, b)  # the bottom half belongs to the dynamics condensation
    finally:
        ctx.close()
