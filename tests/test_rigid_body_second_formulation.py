"""Row (f)3, the pin that can be had without Pinocchio: two independent formulations of the rigid-body dynamics must close on each other.

  first   oracle/rtoc_oracle_rbd.c      recursive Newton-Euler in BODY coordinates (what Robot::RNEA asks of pinocchio::rnea,
                                        include/robotoc/robot/robot.hxx:524-546) + its complex-step derivatives (rtoc_oracle_rbd_cs.c)
  second  oracle/rtoc_oracle_aba.c      articulated-body algorithm (forward dynamics, pinocchio::aba) and composite-rigid-body
                                        algorithm (pinocchio::crba) in WORLD coordinates, own kinematics, own spatial algebra
                                        + its complex-step derivatives (rtoc_oracle_aba_cs.c)

  closure       ID(q, v, FD(q, v, tau, f), f) = tau        i.e.  M a + h = tau with a from the OTHER algorithm
  mass matrix   dID/da (complex step of the first) = M (CRBA of the second)
  derivatives   dID/dq = -M dFD/dq,  dID/dv = -M dFD/dv   (robot.hxx:548-575 is what the device computes analytically)

The GPU's RNEA derivatives are held to the second formulation too (test_gpu_rnea_derivatives_against_the_second_formulation)."""
import numpy as np
import pytest

from robotoc_amd import robot_model as rm


def _case(m, rng, contacts=True):
    q, v, _ = rm.random_configuration(m, rng, 0.8)
    nu = m.nu
    tau = np.concatenate([np.zeros(m.nv - nu), rng.uniform(-20, 20, nu)])
    active = int(rng.integers(0, 1 << m.ncontacts)) if (contacts and m.ncontacts) else 0
    rows = sum((6 if m.contact_type[c] == 1 else 3) for c in range(m.ncontacts) if (active >> c) & 1)
    f = rng.uniform(-30, 30, max(rows, 1))
    return q, v, tau, f, active, rows


@pytest.mark.parametrize("name", ["anymal", "icub", "icub32", "iiwa14"])
def test_forward_dynamics_of_the_second_formulation_closes_the_inverse_dynamics_of_the_first(oracle, name):
    m = rm.load_named(name)
    rng = np.random.default_rng(11)
    worst = 0.0
    for _ in range(6):
        q, v, tau, f, active, rows = _case(m, rng)
        a = oracle.aba_forward_dynamics(m, q, v, tau, f, active)
        # [ID; C] of the first formulation with u = the joint torques: rows [0, nv) are RNEA(q, v, a, f) - [0; u]
        z = np.zeros(3 * m.ncontacts)
        idc = oracle.rbd_eval(m, 0, q, v, a, f, tau[m.nv - m.nu:], active, z)[:m.nv]
        worst = max(worst, np.abs(idc).max() / max(1.0, np.abs(tau).max()))
    print("%s: |ID(q, v, FD(q, v, tau, f), f) - tau| worst %.2e" % (name, worst))
    assert worst < 1e-10


@pytest.mark.parametrize("name", ["anymal", "icub", "iiwa14"])
def test_crba_equals_the_complex_step_of_the_inverse_dynamics_in_the_acceleration(oracle, name):
    m = rm.load_named(name)
    rng = np.random.default_rng(12)
    for _ in range(3):
        q, v, tau, f, active, rows = _case(m, rng, contacts=False)
        M = oracle.aba_crba(m, q)
        z = np.zeros(3 * m.ncontacts)
        _, _, Da = oracle.rbd_linearize_cs(m, 0, q, v, rng.uniform(-1, 1, m.nv), z, tau[m.nv - m.nu:], 0, z)
        assert np.abs(M - M.T).max() < 1e-13 * np.abs(M).max()
        assert np.abs(M - Da[:m.nv]).max() < 1e-12 * np.abs(M).max()
        assert np.abs(M - oracle.rbd_mass_matrix_world(m, q)).max() < 1e-12 * np.abs(M).max()


@pytest.mark.parametrize("name", ["anymal", "icub", "iiwa14"])
def test_rnea_derivatives_follow_from_the_forward_dynamics_derivatives(oracle, name):
    """ID(q, v, FD(q, v, tau)) = tau differentiated: dID/dq + M dFD/dq = 0, dID/dv + M dFD/dv = 0 -- the first formulation's
    complex step on the left, the second formulation's CRBA and complex step on the right."""
    m = rm.load_named(name)
    rng = np.random.default_rng(13)
    worst = 0.0
    for _ in range(3):
        q, v, tau, f, active, rows = _case(m, rng)
        a = oracle.aba_forward_dynamics(m, q, v, tau, f, active)
        M = oracle.aba_crba(m, q)
        dadq, dadv = oracle.aba_linearize_cs(m, q, v, tau, f, active)
        z = np.zeros(3 * m.ncontacts)
        Dq, Dv, _ = oracle.rbd_linearize_cs(m, 0, q, v, a, f, tau[m.nv - m.nu:], active, z)
        for lhs, rhs in ((Dq[:m.nv], -M @ dadq), (Dv[:m.nv], -M @ dadv)):
            worst = max(worst, np.abs(lhs - rhs).max() / max(1.0, np.abs(lhs).max()))
    print("%s: dID/d(q, v) vs -M dFD/d(q, v), worst relative %.2e" % (name, worst))
    assert worst < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", [("anymal", "anymal_trot")])
def test_gpu_rnea_derivatives_against_the_second_formulation(oracle, name, cfg):
    """rtoc_linearize_contact_dynamics' dID/dq, dID/dv, dID/da blocks against -M dFD/dq, -M dFD/dv, M of the articulated-body /
    composite-rigid-body formulation: no step of the comparison touches the recursive Newton-Euler restatement."""
    from robotoc_amd import capi, problems as pr
    from robotoc_amd.types import BUF_CDD, BUF_SOL, GRID_IMPACT
    m = rm.load_named(name)
    dims, grids, _ = getattr(pr, "config_" + cfg)()
    batch = 2
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.set_robot_model(m)
        masks = np.array([(1 << m.ncontacts) - 1 if g.dimf == 3 * m.ncontacts else 0 for g in grids], dtype=np.uint32)
        from test_rigid_body import _masks
        masks = _masks(grids)
        rng = np.random.default_rng(5)
        pos = rng.uniform(-0.5, 0.5, (len(grids), m.ncontacts, 3))
        ctx.set_contact_schedule(masks, pos)
        sol = np.zeros(ctx.shape("sol"))
        o = L.sol.off
        stages = [i for i in range(len(grids) - 1) if grids[i].type != GRID_IMPACT][:6]
        taus = {}
        for b in range(batch):
            for i in range(len(grids)):
                q, v, _ = rm.random_configuration(m, rng, 0.8)
                u = rng.uniform(-5, 5, m.nu)
                f = rng.uniform(-20, 20, 12)
                act = int(masks[i])
                nf = 3 * bin(act).count("1")
                # the acceleration of the iterate is the forward dynamics' answer, so that ID = tau holds at the point of linearisation
                tau = np.concatenate([np.zeros(m.nv - m.nu), u])
                a = oracle.aba_forward_dynamics(m, q, v, tau, f[:max(nf, 1)], act) if i in stages else rng.uniform(-1, 1, m.nv)
                sol[b, i, o[0]:o[0] + m.nq] = q
                sol[b, i, o[1]:o[1] + m.nv] = v
                sol[b, i, o[2]:o[2] + m.nv] = a
                sol[b, i, o[3]:o[3] + m.nu] = u
                sol[b, i, o[4]:o[4] + 12] = f
        ctx.upload(BUF_SOL, sol)
        ctx.linearize_contact_dynamics()
        ctx.sync()
        cdd = ctx.download(BUF_CDD, ctx.shape("cdd"))
        co = L.cdd.off
        nv, ldv = m.nv, dims.nv + dims.nf_max
        worst = dict(dq=0.0, dv=0.0, da=0.0, closure=0.0)
        for b in range(batch):
            for i in stages:
                g, act = grids[i], int(masks[i])
                s = sol[b, i]
                q, v, a = s[o[0]:o[0] + m.nq], s[o[1]:o[1] + nv], s[o[2]:o[2] + nv]
                u, f = s[o[3]:o[3] + m.nu], s[o[4]:o[4] + 12]
                nf = 3 * bin(act).count("1")
                tau = np.concatenate([np.zeros(nv - m.nu), u])
                M2 = oracle.aba_crba(m, q)
                dadq, dadv = oracle.aba_linearize_cs(m, q, v, tau, f[:max(nf, 1)], act)
                rec = cdd[b, i]
                D = rec[co[1]:co[1] + ldv * 2 * nv].reshape(2 * nv, ldv).T
                M = rec[co[0]:co[0] + nv * nv].reshape(nv, nv).T
                idc = rec[co[3]:co[3] + nv]
                sc = lambda x: max(1.0, np.abs(x).max())   # noqa: E731
                worst["closure"] = max(worst["closure"], np.abs(idc).max() / sc(tau))   # the device's ID at a = FD(...) is tau
                worst["da"] = max(worst["da"], np.abs(M - M2).max() / sc(M2))
                worst["dq"] = max(worst["dq"], np.abs(D[:nv, :nv] + M2 @ dadq).max() / sc(D[:nv, :nv]))
                worst["dv"] = max(worst["dv"], np.abs(D[:nv, nv:] + M2 @ dadv).max() / sc(D[:nv, nv:]))
        print("GPU vs the second formulation, worst relative deviation:", worst)
        for k, v_ in worst.items():
            assert v_ < 1e-10, (k, v_)
    finally:
        ctx.close()
