"""linearizeSwitchingConstraint (src/dynamics/switching_constraint.cpp:7-70) on the device, inside rtoc_contact_eval_kkt:
on the grid two ahead of a touch-down the impacting feet must sit at their desired positions at q (+) ((dt1 + dt2) v +
dt1 dt2 a).  Checked against the CPU restatement: P by forward kinematics at the integrated configuration, the Jacobians by
central differences of P (RTOC_OPT_SWITCHING_TRANSPORT = 1, the chain rule) and, for the default -- the reference's
composition of pinocchio::dIntegrateTransport on the transposed Jacobian (robot.hxx:59-92) -- by Pq, dIntegrate_dq and
dIntegrate_dv differenced separately and multiplied the way the reference multiplies them.  Multiplier and STO terms:
the difference between two evaluations with and without xi.  Parity-unpinned (Pinocchio absent)."""
import numpy as np
import pytest

from robotoc_amd import capi, robot_model as rm
from robotoc_amd.grid import ContactSequence, Event, discretize
from robotoc_amd.grid import contact_masks as trot_masks  # the other closed-loop tests import it from here
from robotoc_amd.types import BUF_CDD, BUF_KKT, BUF_SOL, GRID_IMPACT, Records, anymal_dims

Q_STAND = np.array([0, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0])


def q_plus(oracle, m, q, d):
    return oracle.rbd_integrate(m, q, d)


def fd_cols(fun, n, eps=1e-6):
    cols = []
    for k in range(n):
        e = np.zeros(n)
        e[k] = eps
        cols.append((fun(e) - fun(-e)) / (2 * eps))
    return np.stack(cols, axis=1)


Q_ICUB = np.array([0, 0, 0.592, 0, 0, 1, 0,
                   0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726, 0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726,
                   0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0])   # examples/icub/python/jump_sto.py:21-26


@pytest.mark.gpu
@pytest.mark.parametrize("robot,exact", [("anymal", False), ("anymal", True), ("icub", False), ("icub", True)])
def test_switching_constraint_rows_against_the_cpu_restatement(oracle, robot, exact):
    """anymal: two point feet touch down (3 rows each); icub: both soles land after a flight phase (surface contacts:
    6 rows each, P = Log6 of the placement error)"""
    from robotoc_amd.types import icub_dims
    m = rm.load_named(robot)
    N, T, batch = 20, 0.4, 3
    if robot == "anymal":
        dims, q_stand, ncon, ns = anymal_dims(), Q_STAND, 4, 6
        cs = ContactSequence([12, 6, 12], [Event("lift", 0.105), Event("impact", 0.265, impact_dimf=6)])
        phase_masks, impact_masks, imp = [0b1111, 0b1001, 0b1111], [0b0110], [1, 2]
    else:
        dims, q_stand, ncon, ns = icub_dims(35), Q_ICUB, 2, 12
        cs = ContactSequence([12, 0, 12], [Event("lift", 0.105), Event("impact", 0.265, impact_dimf=12)])
        phase_masks, impact_masks, imp = [0b11, 0b00, 0b11], [0b11], [0, 1]
    surface = robot == "icub"
    grids = discretize(N, T, 0.0, cs)
    n, nv, nq, nu = len(grids), m.nv, m.nq, m.nv - 6
    sw = [i for i, g in enumerate(grids) if g.switching_constraint]
    assert len(sw) == 1 and grids[sw[0] + 2].type == GRID_IMPACT and grids[sw[0]].dims == ns
    i0 = sw[0]
    masks = trot_masks(grids, phase_masks, impact_masks)
    rng = np.random.default_rng(5)
    place = [oracle.rbd_contact_placement(m, q_stand, c) for c in range(ncon)]
    pos = np.tile(np.array([p for _, p in place])[None], (n, 1, 1)) + 0.01 * rng.uniform(-1, 1, (n, ncon, 3))
    rot = np.zeros((n, ncon, 3, 3))
    for i in range(n):
        for c in range(ncon):
            rot[i, c] = place[c][0] @ oracle.rbd_exp6(np.concatenate([np.zeros(3), 0.05 * rng.uniform(-1, 1, 3)]))[0]
    ctx = capi.Context(dims, n, batch, 0)
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_contact_schedule(masks, pos, rot.reshape(n, ncon, 9) if surface else None)
    ctx.set_switching_transport(exact)
    wq = np.concatenate([np.full(6, 10.0), np.full(nu, 1.0)])
    ctx.set_configuration_cost(q_stand, np.zeros(nv), np.zeros(nu), wq, np.full(nv, 1.0), np.full(nv, 1e-3), np.full(nu, 1e-3),
                               10.0 * wq, np.full(nv, 1.0), q_weight_impact=wq, v_weight_impact=np.full(nv, 1.0), dv_weight_impact=np.full(nv, 1e-3))
    x0 = np.tile(np.concatenate([q_stand, np.zeros(nv)]), (batch, 1))
    ctx.set_initial_state(x0)
    S, K, D = Records(ctx.L, "sol"), Records(ctx.L, "kkt"), Records(ctx.L, "cdd")
    sol = S.zeros(batch, n)
    for b in range(batch):
        for i in range(n):
            q = q_stand.copy()
            q[:7] = oracle.se3_integrate(q_stand[:7], 0.1 * rng.uniform(-1, 1, 6))
            q[7:] += 0.2 * rng.uniform(-1, 1, nu)
            S.f(sol[b, i], "q")[:nq] = q
            S.f(sol[b, i], "v")[:] = rng.uniform(-1, 1, nv)
            S.f(sol[b, i], "a")[:] = 3.0 * rng.uniform(-1, 1, nv)
            S.f(sol[b, i], "lmd")[:] = rng.uniform(-1, 1, nv)
            S.f(sol[b, i], "gmm")[:] = rng.uniform(-1, 1, nv)
    ctx.upload(BUF_SOL, sol)
    ctx.contact_eval_kkt()
    kkt0, cdd0 = ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd")
    xi = rng.uniform(-1, 1, (batch, ns))
    for b in range(batch):
        S.f(sol[b, i0], "xi")[:ns] = xi[b]
    ctx.upload(BUF_SOL, sol)
    ctx.contact_eval_kkt()
    kkt1, cdd1 = ctx.download_records(BUF_KKT, "kkt"), ctx.download_records(BUF_CDD, "cdd")
    assert (ctx.status() == 0).all()
    dt1, dt2 = grids[i0].dt, grids[i0 + 1].dt
    worst = dict(P=0.0, Phiq=0.0, Phiv=0.0, Phia=0.0, Phit=0.0, lx=0.0, la=0.0, h=0.0, Qtt=0.0, hv=0.0, ha=0.0)
    for b in range(batch):
        s = sol[b, i0]
        q, v, a = S.f(s, "q")[:nq].copy(), S.f(s, "v").copy(), S.f(s, "a").copy()
        dq = (dt1 + dt2) * v + dt1 * dt2 * a
        qp = q_plus(oracle, m, q, dq)

        def P_at(qq):
            if not surface:
                return np.concatenate([oracle.rbd_contact_position(m, qq, c) - pos[i0 + 2, c] for c in imp])
            out = []
            for c in imp:   # Log6(X_desired^-1 X_frame)
                R, p = oracle.rbd_contact_placement(m, qq, c)
                Rd = rot[i0 + 2, c]
                out.append(oracle.rbd_log6(Rd.T @ R, Rd.T @ (p - pos[i0 + 2, c])))
            return np.concatenate(out)

        P = P_at(qp)
        Pq = fd_cols(lambda e: P_at(q_plus(oracle, m, qp, e)), nv)
        if exact:
            Phiq = fd_cols(lambda e: P_at(q_plus(oracle, m, q_plus(oracle, m, q, e), dq)), nv)
            Phiv = fd_cols(lambda e: P_at(q_plus(oracle, m, q, (dt1 + dt2) * (v + e) + dt1 * dt2 * a)), nv)
            Phia = fd_cols(lambda e: P_at(q_plus(oracle, m, q, (dt1 + dt2) * v + dt1 * dt2 * (a + e))), nv)
        else:
            # dIntegrate_dq, dIntegrate_dv on the base (identities on the joints), expressed in the tangent space at q (+) dq
            Tq = fd_cols(lambda e: oracle.se3_difference(qp[:7], oracle.se3_integrate(oracle.se3_integrate(q[:7], e), dq[:6])), 6)
            Tv = fd_cols(lambda e: oracle.se3_difference(qp[:7], oracle.se3_integrate(q[:7], dq[:6] + e)), 6)
            Phiq, Phiv = Pq.copy(), Pq.copy()
            Phiq[:, :6] = Pq[:, :6] @ Tq.T      # Jout^T = dIntegrate Jin^T
            Phiv[:, :6] = Pq[:, :6] @ Tv.T
            Phia = dt1 * dt2 * Phiv
            Phiv = (dt1 + dt2) * Phiv
        k1, c1, k0, c0 = kkt1[b, i0], cdd1[b, i0], kkt0[b, i0], cdd0[b, i0]
        Phix = K.f(k1, "Phix")[:ns]
        worst["P"] = max(worst["P"], np.abs(K.f(k1, "Pres")[:ns] - P).max())
        worst["Phiq"] = max(worst["Phiq"], np.abs(Phix[:, :nv] - Phiq).max())
        worst["Phiv"] = max(worst["Phiv"], np.abs(Phix[:, nv:] - Phiv).max())
        worst["Phia"] = max(worst["Phia"], np.abs(D.f(c1, "Phia")[:ns] - Phia).max())
        Phit = Pq @ (2.0 * (v + dt1 * a))
        worst["Phit"] = max(worst["Phit"], np.abs(K.f(k1, "Phit")[:ns] - Phit).max())
        # multiplier and STO terms (:52-62): what xi adds to the records
        Phix_d, Phia_d = np.array(Phix), np.array(D.f(c1, "Phia")[:ns])
        worst["lx"] = max(worst["lx"], np.abs(K.f(k1, "lx") - K.f(k0, "lx") - Phix_d.T @ xi[b]).max())
        worst["la"] = max(worst["la"], np.abs(D.f(c1, "la") - D.f(c0, "la") - Phia_d.T @ xi[b]).max())
        pqxi = Pq.T @ xi[b]
        worst["h"] = max(worst["h"], abs(K.f(k1, "scal")[2] - K.f(k0, "scal")[2] - xi[b] @ Phit))
        worst["Qtt"] = max(worst["Qtt"], abs(K.f(k1, "scal")[0] - K.f(k0, "scal")[0] - 2.0 * pqxi @ a))
        worst["hv"] = max(worst["hv"], np.abs(K.f(k1, "hx")[nv:] - K.f(k0, "hx")[nv:] - 2.0 * pqxi).max(), np.abs(K.f(k1, "hx")[:nv] - K.f(k0, "hx")[:nv]).max())
        worst["ha"] = max(worst["ha"], np.abs(D.f(c1, "ha") - D.f(c0, "ha") - 2.0 * dt1 * pqxi).max())
        # grids without a switching constraint carry no rows
        for i in range(n):
            if i != i0:
                assert not np.any(K.f(kkt1[b, i], "Phix")) and not np.any(K.f(kkt1[b, i], "Pres"))
                assert np.array_equal(kkt1[b, i], kkt0[b, i])
    print("switching constraint (%s, exact transport %s): worst deviations" % (robot, exact), {k: "%.1e" % e for k, e in worst.items()})
    assert worst["P"] < 1e-12
    assert max(worst[k] for k in ("Phiq", "Phiv", "Phia", "Phit")) < 2e-6   # central differences
    assert max(worst[k] for k in ("lx", "la")) < 1e-10 and max(worst[k] for k in ("h", "hv", "ha", "Qtt")) < 2e-5
    ctx.close()
