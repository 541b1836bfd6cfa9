"""Rigid-body side of evalKKT: rtoc_linearize_contact_dynamics vs the CPU restatement
(oracle/rtoc_oracle_rbd.c), and cross-checks of that restatement itself.

PARITY UNPINNED: the reference delegates these functions to Pinocchio, which is neither vendored nor installed, so no
fixture of the reference's own numbers exists.  What anchors the restatement instead:
  * the model tables come from the reference's test URDFs (tools/urdf_to_model.py), and the reference's own standing pose
    of ANYmal (examples/anymal/*.cpp: base height 0.4792) puts all four feet on the ground to 1e-5 m;
  * the joint-space inertia from the recursion equals a composite-rigid-body sum written in world coordinates;
  * the bias forces satisfy Lagrange's equations (fixed-base arm) and the momentum balance of the floating base,
    both evaluated from kinematics only;
  * the device derivatives (analytical, forward mode) agree with central differences of the restatement.
"""
import os

import numpy as np
import pytest

from robotoc_amd import capi, problems as pr, robot_model as rm
from robotoc_amd.types import BUF_CDD, BUF_KKT, BUF_SOL, GRID_IMPACT, Dims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = rm.MODEL_DIR


def model(name):
    return rm.load_named(name)


def rnea(orc, m, q, v, a):
    z = np.zeros(3 * m.ncontacts)
    return orc.rbd_eval(m, 0, q, v, a, z, np.zeros(m.nu), 0, z)[:m.nv]


def test_anymal_standing_pose_of_the_reference_examples_touches_the_ground(oracle):
    m = model("anymal")
    q = np.array([0, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0])
    feet = np.array([oracle.rbd_contact_position(m, q, c) for c in range(4)])
    assert np.abs(feet[:, 2]).max() < 1e-5
    assert np.allclose(np.abs(feet[:, 0]), abs(feet[0, 0])) and np.allclose(np.abs(feet[:, 1]), abs(feet[0, 1]))
    assert abs(sum(m.mass[i] for i in range(m.njoints)) - 30.4754) < 1e-3


@pytest.mark.parametrize("name", ["anymal", "icub", "iiwa14"])
def test_mass_matrix_from_the_recursion_equals_world_frame_composite_sum(oracle, name):
    m = model(name)
    rng = np.random.default_rng(1)
    for _ in range(3):
        q, v, _ = rm.random_configuration(m, rng)
        z, f = np.zeros(m.nv), np.zeros(3 * m.ncontacts)
        M = np.array([oracle.rbd_eval(m, 1, q, z, e, f, np.zeros(m.nu), 0, f)[:m.nv] for e in np.eye(m.nv)]).T
        Mw = oracle.rbd_mass_matrix_world(m, q)
        assert np.abs(M - Mw).max() < 1e-12 * max(1.0, np.abs(Mw).max())
        assert np.linalg.eigvalsh(Mw).min() > 0
        T, _ = oracle.rbd_energy(m, q, v)
        assert abs(T - 0.5 * v @ Mw @ v) < 1e-12 * max(1.0, T)


def test_lagrange_equations_on_the_fixed_base_arm(oracle):
    """tau = d/dt (dT/dv) - dL/dq with T, U from kinematics only (no force recursion involved)"""
    m = model("iiwa14")
    rng = np.random.default_rng(2)
    h = 1e-5
    for _ in range(3):
        q, v, a = rm.random_configuration(m, rng)
        L = lambda q_: (lambda TU: TU[0] - TU[1])(oracle.rbd_energy(m, q_, v))
        p = lambda q_: oracle.rbd_mass_matrix_world(m, q_) @ v
        dLdq = np.array([(L(q + h * e) - L(q - h * e)) / (2 * h) for e in np.eye(m.nv)])
        dpdt = oracle.rbd_mass_matrix_world(m, q) @ a + sum(((p(q + h * e) - p(q - h * e)) / (2 * h)) * v[j] for j, e in enumerate(np.eye(m.nv)))
        tau = rnea(oracle, m, q, v, a)
        assert np.abs(tau - (dpdt - dLdq)).max() < 1e-6 * max(1.0, np.abs(tau).max())


@pytest.mark.parametrize("name", ["anymal", "icub"])
def test_floating_base_rows_balance_the_rate_of_total_momentum(oracle, name):
    """The base wrench of the recursion = d/dt(total momentum) - gravity wrench, in the base frame."""
    m = model(name)
    rng = np.random.default_rng(3)
    h = 1e-6
    for _ in range(3):
        q, v, a = rm.random_configuration(m, rng)
        mom = lambda s: oracle.rbd_momentum_world(m, oracle.rbd_integrate(m, q, s * v + 0.5 * s * s * a), v + s * a)
        hdot = (mom(h) - mom(-h)) / (2 * h)  # world frame, about the world origin
        # gravity wrench about the world origin
        mass = sum(m.mass[i] for i in range(m.njoints))
        z = np.zeros(m.nv)
        # world com from the potential energy gradient: U = -m g . c
        com = np.zeros(3)
        for k in range(3):
            mk = rm.RobotModel.from_buffer_copy(m)
            mk.gravity[:] = [0.0, 0.0, 0.0]
            mk.gravity[k] = -1.0
            com[k] = oracle.rbd_energy(mk, q, z)[1] / mass
        g = np.array(m.gravity[:])
        wg = np.concatenate([mass * g, np.cross(com, mass * g)])
        w_world = hdot - wg
        # to the base frame: f_b = R^T f, n_b = R^T (n - p x f)
        x, y, zq, w = q[3:7]
        R = np.array([[1 - 2 * (y * y + zq * zq), 2 * (x * y - zq * w), 2 * (x * zq + y * w)],
                      [2 * (x * y + zq * w), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - x * w)],
                      [2 * (x * zq - y * w), 2 * (y * zq + x * w), 1 - 2 * (x * x + y * y)]])
        fb = R.T @ w_world[:3]
        nb = R.T @ (w_world[3:] - np.cross(q[:3], w_world[:3]))
        tau = rnea(oracle, m, q, v, a)
        assert np.abs(tau[:6] - np.concatenate([fb, nb])).max() < 2e-6 * max(1.0, np.abs(tau[:6]).max())


@pytest.mark.parametrize("name", ["anymal", "icub"])
def test_complex_step_derivatives_of_the_restatement(oracle, name):
    """The second witness of the derivative blocks (rtoc_oracle_rbd_cs.c): Im r(x + i h e_j) / h of the restated evaluation.
    Checked here against (1) the world-frame composite mass matrix, which shares nothing with the recursion: dID/da = M(q)
    to 1e-12; (2) central differences of the same evaluation, to their own accuracy."""
    m = model(name)
    rng = np.random.default_rng(11)
    from helpers import check_parity
    wm, wf = 0.0, 0.0
    for trial in range(4):
        q, v, a = rm.random_configuration(m, rng, 0.7)
        nfs = sum(m.contact_rows(c) for c in range(m.ncontacts))
        f, u = rng.uniform(-20, 20, nfs), rng.uniform(-5, 5, m.nu)
        act = int(rng.integers(0, 1 << m.ncontacts))
        pos = rng.uniform(-0.5, 0.5, 3 * m.ncontacts)
        rref = None
        if name == "icub":   # desired placements a moderate twist away from the actual ones (Log6 away from its singularities)
            rref = np.zeros((m.ncontacts, 9))
            for c in range(m.ncontacts):
                Rw, pw = oracle.rbd_contact_placement(m, q, c)
                dR, dp = oracle.rbd_exp6(rng.uniform(-0.3, 0.3, 6))
                rref[c], pos[3 * c:3 * c + 3] = (Rw @ dR).reshape(-1), pw + Rw @ dp
        for impact in (0, 1):
            cs = oracle.rbd_linearize_cs(m, impact, q, v, a, f, u, act, pos, rref=rref)
            fd = oracle.rbd_linearize_fd(m, impact, q, v, a, f, u, act, pos, 1e-6, rref=rref)
            for x, y in zip(fd, cs):
                wf = max(wf, np.abs(x - y).max() / max(1.0, np.abs(y).max()))
            M = oracle.rbd_mass_matrix_world(m, q)
            wm = max(wm, np.abs(cs[2][:m.nv] - M).max() / np.abs(M).max())
    check_parity("complex-step dID/da vs world-frame composite mass matrix", wm, 1e-12)
    check_parity("complex step vs central differences (the differences' accuracy)", wf, 1e-7)


@pytest.mark.parametrize("name", ["anymal"])
def test_restatement_matches_a_recorded_pinocchio_fixture(oracle, name):
    """The pin for f3, the day somebody records it: tools/record_pinocchio_fixture.py writes what Pinocchio computes for the
    calls robotoc::Robot makes (RNEA and its derivatives, the Baumgarte / contact-velocity residuals and theirs, integrate);
    this replays it against the restatement and its complex-step derivatives.  No Pinocchio in this image -> no fixture ->
    skipped, and rtoc_oracle_rbd.c keeps its PARITY UNPINNED header."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "pinocchio_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("no recorded fixture (tools/record_pinocchio_fixture.py needs a machine with Pinocchio)")
    from helpers import check_parity
    fx = np.load(path, allow_pickle=True)
    m = model(name)
    assert [str(x) for x in fx["joint_names"]] == rm.joint_names(name), "joint order of the model table differs from Pinocchio's"
    worst = dict(ID=0.0, C=0.0, dID=0.0, dC=0.0, integrate=0.0)
    sc = lambda x: max(1.0, np.abs(x).max()) if np.size(x) else 1.0
    for k in range(len(fx["q"])):
        q, v, a, f, act, impact, pos = fx["q"][k], fx["v"][k], fx["a"][k], fx["f"][k], int(fx["active"][k]), int(fx["impact"][k]), fx["pos"][k]
        u = np.zeros(m.nu)
        val = oracle.rbd_eval(m, impact, q, v, a, f, u, act, pos)
        Dq, Dv, Da = oracle.rbd_linearize_cs(m, impact, q, v, a, f, u, act, pos)
        nv = m.nv
        worst["ID"] = max(worst["ID"], np.abs(val[:nv] - fx["ID"][k]).max() / sc(fx["ID"][k]))   # u = 0: ID is the RNEA torque
        worst["dID"] = max(worst["dID"], np.abs(Dq[:nv] - fx["dIDdq"][k]).max() / sc(fx["dIDdq"][k]), np.abs(Da[:nv] - fx["dIDda"][k]).max() / sc(fx["dIDda"][k]))
        if not impact:
            worst["dID"] = max(worst["dID"], np.abs(Dv[:nv] - fx["dIDdv"][k]).max() / sc(fx["dIDdv"][k]))
        C = np.asarray(fx["C"][k], dtype=float)
        if C.size:
            worst["C"] = max(worst["C"], np.abs(val[nv:] - C).max() / sc(C))
            for D, key in ((Dq, "dCdq"), (Dv, "dCdv")) + (() if impact else ((Da, "dCda"),)):
                ref = np.asarray(fx[key][k], dtype=float)
                worst["dC"] = max(worst["dC"], np.abs(D[nv:] - ref).max() / sc(ref))
        worst["integrate"] = max(worst["integrate"], np.abs(oracle.rbd_integrate(m, q, fx["dq"][k]) - fx["q_plus_dq"][k]).max())
    for key, w in worst.items():
        check_parity("restatement vs Pinocchio: " + key, w, 1e-10)


def test_log6_inverts_exp6(oracle):
    """pinocchio::log6 restated in the oracle against the SE(3) exponential of orc_rbd_integrate: log6(exp6(xi)) = xi"""
    rng = np.random.default_rng(9)
    for scale in (1e-9, 1e-5, 1e-2, 0.5, 1.5):  # |w| < pi
        for _ in range(20):
            xi = rng.uniform(-1, 1, 6) * scale
            R, p = oracle.rbd_exp6(xi)
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-14
            assert np.abs(oracle.rbd_log6(R, p) - xi).max() < 1e-9 * max(1.0, scale) + 1e-15


@pytest.mark.parametrize("name", ["anymal", "icub"])
def test_host_side_frame_placement_and_contact_masks(oracle, name):
    """RobotModel.frame_placement (problem set-up on the host: contact positions, initial contact forces) against the
    restatement's kinematics; contact_masks against the phase structure of the trot grid"""
    m = rm.load_named(name)
    rng = np.random.default_rng(2)
    for _ in range(5):
        q = rm.random_configuration(m, rng)[0]
        for k in range(m.ncontacts):
            R, p = m.frame_placement(q, k)
            R2, p2 = oracle.rbd_contact_placement(m, q, k)
            assert np.abs(R - R2).max() < 1e-14 and np.abs(p - p2).max() < 1e-14
    from robotoc_amd.grid import ANYMAL_TROT_IMPACT_MASKS, ANYMAL_TROT_PHASE_MASKS, contact_masks
    _, grids, _ = pr.config_anymal_trot()
    masks = contact_masks(grids, ANYMAL_TROT_PHASE_MASKS, ANYMAL_TROT_IMPACT_MASKS)
    assert all(3 * bin(int(k)).count("1") == g.dimf for k, g in zip(masks, grids))
    for i, g in enumerate(grids):
        if g.switching_constraint:   # the feet that touch down two grid points ahead are in the air now
            assert grids[i + 2].type == GRID_IMPACT and int(masks[i]) & int(masks[i + 2]) == 0 and 3 * bin(int(masks[i + 2])).count("1") == g.dims


def test_model_table_is_validated():
    capi.build()
    lib = capi.lib()
    assert hasattr(lib, "rtoc_set_robot_model") and hasattr(lib, "rtoc_linearize_contact_dynamics")


def test_tangent_walk_plans_of_the_bundled_robots():
    """rtoc_robot_model_plan (host arithmetic, no device): what rtoc_set_robot_model plans for the tangent walk of
    rtoc_linearize_contact_dynamics -- LDS slots for forward tangents = branching joints on a path (ANYmal: the base; iCub: base
    and chest; a chain: none), dofs per pass chosen per model, the bodies a pass visits = ancestors and subtrees of its dofs, and
    the LDS bytes per wave that decide how many grid points a CU holds (ANYmal: 20,000 B = eight waves per CU)."""
    plan, bodies = capi.robot_model_plan(model("anymal"))
    assert (plan.nlevels, plan.nbranch, plan.dofs_per_pass, plan.npass, plan.lds_bytes) == (4, 1, 18, 1, 20000)
    assert bodies == [list(range(13))]
    assert 8 * ((plan.lds_bytes + 1279) // 1280 * 1280) <= 160 * 1024
    # two passes of 9 dofs: the second (RF, LH, RH legs... dofs 9..17) does not visit the leg whose dofs all sit in the first
    plan, bodies = capi.robot_model_plan(model("anymal"), 9)
    assert plan.npass == 2 and bodies[0] == list(range(13)) and bodies[1] == [0] + list(range(4, 13))
    m = model("icub")
    plan, bodies = capi.robot_model_plan(m)
    assert (plan.nlevels, plan.nbranch, plan.dofs_per_pass, plan.npass) == (11, 2, 19, 2)
    assert bodies[0] == list(range(m.njoints))
    # dofs 19..34: the second half of the right leg, the torso and the arms -- no body of the left leg, none of the right leg above
    par = list(m.parent[:m.njoints])
    dof_body = {m.idx_v[i] + k: i for i in range(m.njoints) for k in range(6 if i == 0 and m.floating_base else 1)}
    want = set()
    for j in range(19, m.nv):
        b = dof_body[j]
        k = b
        while k >= 0:              # ancestors
            want.add(k)
            k = par[k]
        want |= {i for i in range(m.njoints) if b in _ancestors(par, i)}   # subtree
    assert bodies[1] == sorted(want) and len(bodies[1]) < m.njoints
    plan, _ = capi.robot_model_plan(model("icub32"))
    assert (plan.nlevels, plan.nbranch, plan.dofs_per_pass, plan.npass) == (8, 1, 21, 2)
    plan, bodies = capi.robot_model_plan(model("iiwa14"))
    assert (plan.nlevels, plan.nbranch, plan.dofs_per_pass, plan.npass) == (7, 0, 7, 1) and bodies == [list(range(7))]
    # a table whose joints do not come depth first is refused here as by rtoc_set_robot_model
    a = model("anymal")
    bad = type(a).from_buffer_copy(a)
    bad.parent[5] = 3
    with pytest.raises(capi.RtocError):
        capi.robot_model_plan(bad)
    with pytest.raises(capi.RtocError):
        capi.robot_model_plan(a, 22)


def _ancestors(par, i):
    out = []
    while i >= 0:
        out.append(i)
        i = par[i]
    return out


@pytest.mark.gpu
def test_model_joints_must_come_depth_first():
    """rtoc_set_robot_model: the walk keeps one value block per OPEN tree level, so when a joint is visited its parent has to be
    the joint open one level up.  A table whose joint 5 hangs under joint 3 AFTER joint 4 (a sibling of joint 1) has closed that
    branch is refused (the check used to look only at the level of the parent, which joint 3 still occupied)."""
    m = model("anymal")
    ctx = capi.Context(pr.config_anymal_trot()[0], 4, 1, 0)
    try:
        ctx.set_robot_model(m)
        bad = type(m).from_buffer_copy(m)
        assert list(bad.parent[:6]) == [-1, 0, 1, 2, 0, 4]
        bad.parent[5] = 3
        with pytest.raises(capi.RtocError):
            ctx.set_robot_model(bad)
    finally:
        ctx.close()


def _masks(grids):
    """contact masks with popcount * 3 == dimf: all four feet, or alternating diagonal pairs"""
    out, flip = [], False
    for g in grids:
        if g.dimf == 12:
            out.append(0b1111)
        elif g.dimf == 6:
            out.append(0b0110 if flip else 0b1001)
            flip = not flip
        elif g.dimf == 0:
            out.append(0)
        else:
            raise AssertionError(g.dimf)
    return np.array(out, dtype=np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("cfg", ["anymal_trot", "anymal_jump_sto"])
def test_gpu_linearisation_matches_the_restatement_and_its_complex_step_derivatives(oracle, cfg, fused):
    m = model("anymal")
    dims, grids, _ = getattr(pr, "config_" + cfg)()
    batch = 3
    ctx = capi.Context(dims, len(grids), batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_linearize_fused(fused)   # one kernel, or values pre-pass + tangent walk (the default)
    masks = _masks(grids)
    rng = np.random.default_rng(4)
    pos = rng.uniform(-0.5, 0.5, (len(grids), 4, 3))
    ctx.set_contact_schedule(masks, pos)
    sol = np.zeros(ctx.shape("sol"))
    o = L.sol.off
    for b in range(batch):
        for i in range(len(grids)):
            q, v, a = rm.random_configuration(m, rng, 0.8)
            sol[b, i, o[0]:o[0] + m.nq] = q
            sol[b, i, o[1]:o[1] + m.nv] = v
            sol[b, i, o[2]:o[2] + m.nv] = a
            sol[b, i, o[3]:o[3] + m.nu] = rng.uniform(-5, 5, m.nu)
            sol[b, i, o[4]:o[4] + 12] = rng.uniform(-20, 20, 12)
    ctx.upload(BUF_SOL, sol)
    ctx.linearize_contact_dynamics()
    ctx.sync()
    cdd = ctx.download(BUF_CDD, ctx.shape("cdd"))
    co = L.cdd.off
    nv, ldv, nfm = m.nv, dims.nv + dims.nf_max, dims.nf_max
    worst = dict(val=0.0, dq=0.0, dv=0.0, da=0.0)
    for b in range(batch):
        for i in range(len(grids) - 1):
            g, act = grids[i], int(masks[i])
            impact = g.type == GRID_IMPACT
            n = nv + g.dimf
            s = sol[b, i]
            q, v, a = s[o[0]:o[0] + m.nq], s[o[1]:o[1] + nv], s[o[2]:o[2] + nv]
            u, f = s[o[3]:o[3] + m.nu], s[o[4]:o[4] + 12]
            ref = oracle.rbd_eval(m, impact, q, v, a, f, u, act, pos[i].reshape(-1))
            # complex-step derivatives of the restated evaluation: exact to rounding (rtoc_oracle_rbd_cs.c), where central
            # differences stop at ~2e-9
            Dq, Dv, Da = oracle.rbd_linearize_cs(m, impact, q, v, a, f, u, act, pos[i].reshape(-1))
            rec = cdd[b, i]
            idc = rec[co[3]:co[3] + n]
            D = rec[co[1]:co[1] + ldv * 2 * nv].reshape(2 * nv, ldv).T  # [row, col]
            M = rec[co[0]:co[0] + nv * nv].reshape(nv, nv).T
            J = rec[co[2]:co[2] + nfm * nv].reshape(nv, nfm).T[:g.dimf]
            scale = lambda x: max(1.0, np.abs(x).max())
            worst["val"] = max(worst["val"], np.abs(idc - ref).max() / scale(ref))
            worst["dq"] = max(worst["dq"], np.abs(D[:n, :nv] - Dq).max() / scale(Dq))
            worst["da"] = max(worst["da"], np.abs(M - Da[:nv]).max() / scale(Da))
            assert np.abs(M - M.T).max() < 1e-12 * scale(M)
            if g.dimf == 0:
                worst["dv"] = max(worst["dv"], np.abs(D[:n, nv:] - Dv).max() / scale(Dv))
            elif impact:
                # the v block: only the contact-velocity rows are defined (= dC/d(dv))
                worst["dv"] = max(worst["dv"], np.abs(D[nv:n, nv:] - Dv[nv:]).max() / scale(Dv))
                worst["dv"] = max(worst["dv"], np.abs(J - Da[nv:]).max() / scale(Da))
            else:
                worst["dv"] = max(worst["dv"], np.abs(D[:n, nv:] - Dv).max() / scale(Dv))
                worst["da"] = max(worst["da"], np.abs(J - Da[nv:]).max() / scale(Da))
    print("worst relative deviation:", worst)
    from helpers import check_parity
    check_parity("ID, C values vs restatement", worst["val"], 1e-14)   # observed 4e-16 on the MI355X
    for k in ("dq", "dv", "da"):   # observed 1e-15 on the MI355X
        check_parity("d[ID; C]/d%s vs complex step" % k[1], worst[k], 1e-13)
    # ---- multiplier terms (contact_dynamics.cpp:35-52, impact_dynamics.cpp:19-27) against the same algebra in numpy,
    # on the derivative blocks the device just produced ----
    kkt0 = rng.uniform(-1, 1, ctx.shape("kkt"))
    cdd0 = cdd.copy()
    for fld, size in ((7, nv), (8, nfm), (12, 8)):  # RTOC_CDD_LA, LF, LUP
        cdd0[:, :, co[fld]:co[fld] + size] = rng.uniform(-1, 1, (batch, len(grids), size))
    for b in range(batch):
        for i in range(len(grids)):
            sol[b, i, o[7]:o[7] + nv] = rng.uniform(-1, 1, nv)     # beta
            sol[b, i, o[8]:o[8] + 12] = rng.uniform(-1, 1, 12)     # mu_stack
            sol[b, i, o[9]:o[9] + 6] = rng.uniform(-1, 1, 6)       # nu_passive
    ctx.upload(BUF_SOL, sol)
    ctx.upload(BUF_KKT, kkt0)
    ctx.upload(BUF_CDD, cdd0)
    ctx.linearize_contact_dynamics(augment_residual=True)
    ctx.sync()
    kkt1, cdd1 = ctx.download(BUF_KKT, ctx.shape("kkt")), ctx.download(BUF_CDD, ctx.shape("cdd"))
    ko = L.kkt.off
    KKT_LX, KKT_LU = 6, 7  # RTOC_KKT_LX, RTOC_KKT_LU (include/rtoc_layout.h)
    wa = 0.0
    for b in range(batch):
        for i in range(len(grids) - 1):
            g = grids[i]
            impact = g.type == GRID_IMPACT
            nf, n = g.dimf, nv + g.dimf
            s = sol[b, i]
            beta, mu, nup = s[o[7]:o[7] + nv], s[o[8]:o[8] + nf], s[o[9]:o[9] + 6]
            rec = cdd1[b, i]
            D = rec[co[1]:co[1] + ldv * 2 * nv].reshape(2 * nv, ldv).T
            M = rec[co[0]:co[0] + nv * nv].reshape(nv, nv).T
            J = rec[co[2]:co[2] + nfm * nv].reshape(nv, nfm).T[:nf]
            w = np.concatenate([beta, mu])
            lx0, lx1 = kkt0[b, i, ko[KKT_LX]:ko[KKT_LX] + 2 * nv], kkt1[b, i, ko[KKT_LX]:ko[KKT_LX] + 2 * nv]
            la0, la1 = cdd0[b, i, co[7]:co[7] + nv], rec[co[7]:co[7] + nv]
            lf0, lf1 = cdd0[b, i, co[8]:co[8] + nf], rec[co[8]:co[8] + nf]
            lu0, lu1 = kkt0[b, i, ko[KKT_LU]:ko[KKT_LU] + m.nu], kkt1[b, i, ko[KKT_LU]:ko[KKT_LU] + m.nu]
            if impact:
                exp_lq = lx0[:nv] + D[:n, :nv].T @ w
                exp_lv = lx0[nv:] + D[nv:n, nv:].T @ mu
                exp_la = la0 + M.T @ beta + D[nv:n, nv:].T @ mu
                exp_lf = lf0 - D[nv:n, nv:] @ beta
                exp_lu, exp_lup = lu0, np.zeros(6)
            else:
                exp_lq = lx0[:nv] + D[:n, :nv].T @ w
                exp_lv = lx0[nv:] + D[:n, nv:].T @ w
                exp_la = la0 + M.T @ beta + J.T @ mu
                exp_lf = lf0 - J @ beta
                exp_lu, exp_lup = lu0 - beta[6:], nup - beta[:6]
            for got, exp in ((lx1[:nv], exp_lq), (lx1[nv:], exp_lv), (la1, exp_la), (lf1, exp_lf), (lu1, exp_lu), (rec[co[12]:co[12] + 6], exp_lup)):
                if exp.size:
                    wa = max(wa, np.abs(got - exp).max() / max(1.0, np.abs(exp).max()))
    print("multiplier terms, worst relative deviation:", wa)
    assert wa < 1e-12
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dpp", [9, 6, 5])
def test_gpu_linearisation_does_not_depend_on_the_passes(dpp):
    """ANYmal trot (ordinary, impact and flight grids): the linearisation and its multiplier terms with 2, 3 and 4 passes
    (RTOC_OPT_LINEARIZE_DOFS_PER_PASS; passes behind the first skip the legs none of their dofs sits in) against the single pass
    of 18 dofs the library chooses; NaN-filled records beforehand: every entry the passes skip is written as zero."""
    from robotoc_amd.types import OPT_LINEARIZE_DOFS_PER_PASS, Records
    m = model("anymal")
    dims, grids, _ = pr.config_anymal_trot()
    batch = 2
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        Rc = Records(L, "cdd")
        ctx.set_grid(grids)
        ctx.set_robot_model(m)
        assert ctx.get_option(OPT_LINEARIZE_DOFS_PER_PASS) == 18
        rng = np.random.default_rng(14)
        ctx.set_contact_schedule(_masks(grids), rng.uniform(-0.5, 0.5, (len(grids), 4, 3)))
        sol = np.zeros(ctx.shape("sol"))
        o = L.sol.off
        for b in range(batch):
            for i in range(len(grids)):
                q, v, a = rm.random_configuration(m, rng, 0.8)
                sol[b, i, o[0]:o[0] + m.nq], sol[b, i, o[1]:o[1] + m.nv], sol[b, i, o[2]:o[2] + m.nv] = q, v, a
                sol[b, i, o[3]:o[3] + m.nu] = rng.uniform(-5, 5, m.nu)
                sol[b, i, o[4]:o[4] + 12] = rng.uniform(-20, 20, 12)
                sol[b, i, o[7]:o[7] + m.nv] = rng.uniform(-1, 1, m.nv)
                sol[b, i, o[8]:o[8] + 12] = rng.uniform(-1, 1, 12)
        ctx.upload(BUF_SOL, sol)
        out = []
        for d in (0, dpp):
            ctx.set_linearize_dofs_per_pass(d)
            assert ctx.get_option(OPT_LINEARIZE_DOFS_PER_PASS) == (d if d else 18)
            ctx.upload(BUF_KKT, np.zeros(ctx.shape("kkt")))
            cdd0 = np.full(ctx.shape("cdd"), np.nan)
            for f in ("la", "lf"):   # the residuals the multiplier terms are accumulated into
                Rc.f(cdd0, f)[...] = 0.0
            ctx.upload(BUF_CDD, cdd0)
            ctx.linearize_contact_dynamics(True)
            ctx.sync()
            out.append((ctx.download(BUF_KKT, ctx.shape("kkt")), ctx.download(BUF_CDD, ctx.shape("cdd"))))
            assert (ctx.status() == 0).all()
        for f in ("IDC", "dIDda", "dCda", "dIDCdqv", "la", "lf"):
            a0, a1 = Rc.f(out[0][1], f)[:, :-1], Rc.f(out[1][1], f)[:, :-1]
            assert np.array_equal(np.isnan(a0), np.isnan(a1)), f          # the same entries written
            ok = ~np.isnan(a0)
            assert ok.any() and np.abs(a0[ok] - a1[ok]).max() <= 1e-13 * max(1.0, np.abs(a0[ok]).max()), f
        assert np.abs(out[0][0] - out[1][0]).max() <= 1e-13 * max(1.0, np.abs(out[0][0]).max())
        assert np.abs(out[0][0]).max() > 1e-3
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused,dpp", [(False, 0), (True, 0), (False, 21), (False, 12), (False, 7), (True, 12)])
def test_gpu_linearisation_icub_surface_contacts_two_passes_eleven_levels(oracle, fused, dpp):
    """nv = 35: several passes over 11 tree levels (RTOC_OPT_LINEARIZE_DOFS_PER_PASS; 0: the library's choice, 19 dofs per pass;
    a pass visits only the bodies its dofs can move or load and writes zeros for the others -- the records are filled with NaN
    beforehand); two SURFACE contacts (the soles: 6 rows each, wrench in the local frame, Log6 position / orientation error
    against a desired placement).  The multiplier terms must not depend on the split into passes."""
    from robotoc_amd.types import Grid, GRID_INTERMEDIATE, GRID_TERMINAL
    m = model("icub")
    assert m.contact_rows(0) == 6 and m.max_dimf == 12
    dims = Dims(35, 29, 6, 12, 12, 0)
    inter = lambda dimf: Grid(GRID_INTERMEDIATE, 0, 0, 0, dimf, 0, 8, 2, 0.02)
    grids = [inter(6), inter(12), Grid(GRID_IMPACT, 0, 0, 0, 6, 0, 1, -1, 0.0), inter(12), inter(0), Grid(GRID_TERMINAL, 0, 0, 0, 0, 0, 1, 2, 0.0)]
    masks = np.array([0b10, 0b11, 0b01, 0b11, 0, 0], dtype=np.uint32)
    batch = 2
    ctx = capi.Context(dims, len(grids), batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    ctx.set_robot_model(m)
    ctx.set_linearize_fused(fused)
    ctx.set_linearize_dofs_per_pass(dpp)
    from robotoc_amd.types import OPT_LINEARIZE_DOFS_PER_PASS
    assert ctx.get_option(OPT_LINEARIZE_DOFS_PER_PASS) == (dpp if dpp else 19)
    rng = np.random.default_rng(5)
    # one configuration per grid point (the schedule is shared by the instances); the desired placements are the actual
    # ones moved by a twist of up to ~0.3 (a contact that drifted: Log6 stays away from its singularity at pi)
    qs = [rm.random_configuration(m, rng, 0.6)[0] for _ in grids]
    pos, rot = np.zeros((len(grids), 2, 3)), np.zeros((len(grids), 2, 3, 3))
    for i in range(len(grids)):
        for c in range(2):
            Rw, pw = oracle.rbd_contact_placement(m, qs[i], c)
            dR, dp = oracle.rbd_exp6(rng.uniform(-0.3, 0.3, 6))
            rot[i, c], pos[i, c] = Rw @ dR, pw + Rw @ dp
    ctx.set_contact_schedule(masks, pos, rot)
    sol = np.zeros(ctx.shape("sol"))
    o, co = L.sol.off, L.cdd.off
    for b in range(batch):
        for i in range(len(grids)):
            _, v, a = rm.random_configuration(m, rng, 0.6)
            sol[b, i, o[0]:o[0] + m.nq], sol[b, i, o[1]:o[1] + m.nv], sol[b, i, o[2]:o[2] + m.nv] = qs[i], v, a
            sol[b, i, o[3]:o[3] + m.nu] = rng.uniform(-5, 5, m.nu)
            sol[b, i, o[4]:o[4] + 12] = rng.uniform(-20, 20, 12)
            sol[b, i, o[7]:o[7] + m.nv] = rng.uniform(-1, 1, m.nv)     # beta, mu: the multiplier terms below
            sol[b, i, o[8]:o[8] + 12] = rng.uniform(-1, 1, 12)
    ctx.upload(BUF_SOL, sol)
    # the multiplier terms (accumulated into zeroed residuals) with this split into passes and with one of 21 dofs per pass
    aug = []
    for d in (dpp, 21):
        ctx.set_linearize_dofs_per_pass(d)
        ctx.upload(BUF_KKT, np.zeros(ctx.shape("kkt")))
        ctx.upload(BUF_CDD, np.zeros(ctx.shape("cdd")))
        ctx.linearize_contact_dynamics(True)
        ctx.sync()
        aug.append((ctx.download(BUF_KKT, ctx.shape("kkt")), ctx.download(BUF_CDD, ctx.shape("cdd"))))
    for x, y in zip(*aug):
        assert np.abs(x - y).max() <= 1e-13 * max(1.0, np.abs(y).max())
    assert np.abs(aug[0][0]).max() > 1e-3
    ctx.set_linearize_dofs_per_pass(dpp)
    ctx.upload(BUF_CDD, np.full(ctx.shape("cdd"), np.nan))
    ctx.linearize_contact_dynamics()
    ctx.sync()
    cdd = ctx.download(BUF_CDD, ctx.shape("cdd"))
    nv, ldv, nfm = m.nv, dims.nv + dims.nf_max, dims.nf_max
    worst = 0.0
    for b in range(batch):
        for i in range(len(grids) - 1):
            g, act = grids[i], int(masks[i])
            impact = g.type == GRID_IMPACT
            n = nv + g.dimf
            s = sol[b, i]
            args = (m, impact, s[o[0]:o[0] + m.nq], s[o[1]:o[1] + nv], s[o[2]:o[2] + nv], s[o[4]:o[4] + 12], s[o[3]:o[3] + m.nu], act, pos[i].reshape(-1))
            ref = oracle.rbd_eval(*args, rref=rot[i].reshape(2, 9))
            Dq, Dv, Da = oracle.rbd_linearize_cs(*args, rref=rot[i].reshape(2, 9))
            rec = cdd[b, i]
            D = rec[co[1]:co[1] + ldv * 2 * nv].reshape(2 * nv, ldv).T
            M = rec[co[0]:co[0] + nv * nv].reshape(nv, nv).T
            J = rec[co[2]:co[2] + nfm * nv].reshape(nv, nfm).T[:g.dimf]
            assert np.abs(rec[co[3]:co[3] + n] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (b, i)
            # every entry the condensation reads has been written (zeros included), whatever the passes skipped
            assert np.isfinite(D[:n, :nv]).all() and np.isfinite(M).all() and np.isfinite(J).all(), (b, i)
            assert np.isfinite(D[nv:n, nv:] if impact else D[:n, nv:]).all(), (b, i)
            sc = lambda x: max(1.0, np.abs(x).max())
            worst = max(worst, np.abs(D[:n, :nv] - Dq).max() / sc(Dq), np.abs(M - Da[:nv]).max() / sc(Da))
            if impact:
                worst = max(worst, np.abs(D[nv:n, nv:] - Dv[nv:]).max() / sc(Dv), np.abs(J - Da[nv:]).max() / sc(Da))
            else:
                worst = max(worst, np.abs(D[:n, nv:] - Dv).max() / sc(Dv))
                if g.dimf:
                    worst = max(worst, np.abs(J - Da[nv:]).max() / sc(Da))
    print("iCub (surface contacts) worst relative deviation of the derivatives:", worst)
    from helpers import check_parity
    check_parity("iCub d[ID; C]/d(q, v, a) vs complex step", worst, 1e-13)   # observed 2.6e-15 on the MI355X
    ctx.close()


def test_icub32_is_the_icub_with_the_torso_locked(oracle):
    """robot_model.lock_joints: the nv = 32 iCub (BASELINE.json's size) is the reference URDF's nv = 35 model with the three torso
    joints welded at angle zero.  Inverse dynamics, contact residuals and contact-frame placements of the two models agree when the
    full model's torso joints rest (q = v = a = 0 there) -- on the rows of the joints both have; the mass is conserved."""
    full, red = rm.load_named("icub"), rm.load_named("icub32")
    assert (full.nv, red.nv, red.nq, red.njoints, red.ncontacts) == (35, 32, 33, 27, 2)
    names = rm.joint_names("icub")
    locked = [i for i, n in enumerate(names) if n in rm.ICUB32_LOCKED]
    keep_v = [k for k in range(full.nv) if k not in [full.idx_v[i] for i in locked]]
    keep_q = [k for k in range(full.nq) if k not in [full.idx_q[i] for i in locked]]
    assert abs(sum(full.mass[i] for i in range(full.njoints)) - sum(red.mass[i] for i in range(red.njoints))) < 1e-12
    rng = np.random.default_rng(3)
    for trial in range(3):
        q, v, a = rm.random_configuration(full, rng, 0.7)
        for i in locked:
            q[full.idx_q[i]] = v[full.idx_v[i]] = a[full.idx_v[i]] = 0.0
        f = rng.uniform(-30, 30, 12)
        pref, rref = rng.uniform(-0.3, 0.3, (2, 3)), np.tile(np.eye(3).reshape(9), (2, 1))
        u_full = rng.uniform(-5, 5, full.nv - 6)
        u_red = np.array([u_full[k - 6] for k in keep_v[6:]])
        r_full = oracle.rbd_eval(full, False, q, v, a, f, u_full, 0b11, pref, rref)
        r_red = oracle.rbd_eval(red, False, q[keep_q], v[keep_v], a[keep_v], f, u_red, 0b11, pref, rref)
        assert np.abs(r_full[keep_v] - r_red[:32]).max() < 1e-9 * max(1.0, np.abs(r_full).max())   # ID rows of the common joints
        assert np.abs(r_full[35:] - r_red[32:]).max() < 1e-10                                      # contact rows
        for c in range(2):
            Rf, pf = oracle.rbd_contact_placement(full, q, c)
            Rr, pr_ = oracle.rbd_contact_placement(red, q[keep_q], c)
            assert np.abs(Rf - Rr).max() < 1e-12 and np.abs(pf - pr_).max() < 1e-12
