"""Regenerates tests/golden/ref_*.npz: small seeded problems with the outputs of the REFERENCE'S OWN SOURCES
(oracle/_ref/librtoc_ref.so: robotoc's src/riccati, src/dynamics, src/core compiled from /root/reference against
oracle/ref_shim; see oracle/ref.py).  These are the fixtures that pin the oracle (CPU test) and the HIP path (GPU test,
no oracle and no reference in the loop) to what robotoc's code computes.  Needs /root/reference:

  python tests/golden/make_ref_golden.py        (from the repo root)
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, ref  # noqa: E402
from robotoc_amd import problems as pr  # noqa: E402
from robotoc_amd.grid import anymal_trot_sequence, discretize, jump_sto_sequence  # noqa: E402
from robotoc_amd.types import GRID_TERMINAL, Records, anymal_dims  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def grid_table(grids):
    return dict(grid=np.array([[g.type, g.sto, g.sto_next, g.switching_constraint, g.dimf, g.dims, g.num_grids_in_phase,
                                g.time_stage] for g in grids]), grid_dt=np.array([g.dt for g in grids]))


def anymal_trot_short(N=8):
    """ANYmal trot cut short: one lift, one impact, one switching-constraint grid."""
    cs = anymal_trot_sequence(t0=0.03, swing=0.05, double_support=0.03, cycles=1)
    return anymal_dims(), discretize(N, N * 0.02, 0.0, cs)


def anymal_jump_sto_short(N=8):
    """ANYmal jump with switching-time optimisation cut short: lift + impact with sto, ns = 12."""
    cs = jump_sto_sequence(ground_time=0.05, flying_time=0.06, nf=12)
    return anymal_dims(), discretize(N, N * 0.02, 0.0, cs, phase_based=True)


def riccati_fixture(name, dims, grids, mode):
    L = orc.layout(dims)
    kkt = pr.make_kkt_batch(L, grids, 1, mode=mode)
    dx0 = pr.make_dx0(L, 1)
    ric, d = Records(L, "ric").zeros(1, len(grids)), Records(L, "dir").zeros(1, len(grids))
    Records(L, "dir").f(d[0, 0], "dx")[...] = dx0[0]
    k = kkt[0].copy()
    ref.riccati_sweep(L, grids, k, ric[0], d[0])
    np.savez_compressed(os.path.join(HERE, name), kkt=kkt, dx0=dx0, ric=ric, dir=d, kkt_mutated=k[None], **grid_table(grids))


def riccati_fixture_full_size(name, dims, grids, mode):
    """The BASELINE configurations at their full sizes (N = 40 / N = 30): the outputs of the reference's sources only -- the
    inputs come from the seeded generator of robotoc_amd/problems.py, which the GPU box has too; their SHA-256 is stored so
    that a drift of the generator is noticed instead of silently comparing other problems."""
    import hashlib
    L = orc.layout(dims)
    kkt = pr.make_kkt_batch(L, grids, 1, mode=mode)
    dx0 = pr.make_dx0(L, 1)
    ric, d = Records(L, "ric").zeros(1, len(grids)), Records(L, "dir").zeros(1, len(grids))
    Records(L, "dir").f(d[0, 0], "dx")[...] = dx0[0]
    ref.riccati_sweep(L, grids, kkt[0].copy(), ric[0], d[0])
    sha = hashlib.sha256(np.ascontiguousarray(kkt).tobytes() + np.ascontiguousarray(dx0).tobytes()).hexdigest()
    np.savez_compressed(os.path.join(HERE, name), ric=ric, dir=d, inputs_sha256=np.array(sha), mode=np.array(mode),
                        dims=np.array([dims.nv, dims.nu, dims.np, dims.nf_max, dims.ns_max, dims.nc_max]), **grid_table(grids))


def full_size_fixtures():
    riccati_fixture_full_size("ref_anymal_trot_n40_riccati.npz", *pr.config_anymal_trot()[:2], mode="dynamics")
    riccati_fixture_full_size("ref_anymal_jump_sto_n40_riccati.npz", *pr.config_anymal_jump_sto()[:2], mode="dynamics")
    riccati_fixture_full_size("ref_icub35_jump_n30_riccati.npz", *pr.config_icub_jump(nv=35)[:2], mode="factory")
    icub32_full_size()


def icub32_full_size():
    """BASELINE configs[3] at the size BASELINE.json names (nv = 32, N = 30)"""
    riccati_fixture_full_size("ref_icub32_jump_n30_riccati.npz", *pr.config_icub_jump(nv=32)[:2], mode="factory")


def condense_fixture(name, dims, grids):
    """Per grid point: condenseContactDynamics / condenseImpactDynamics, then the expansions on seeded directions.
    num_grids_in_phase = 1 everywhere, so that the evalKKT-tail scalings (not part of these reference functions) are
    the identity for whoever reproduces the fixture through rtoc_condense."""
    L = orc.layout(dims)
    grids = [copy.copy(g) for g in grids]
    for g in grids:
        g.num_grids_in_phase = 1
    kkt, cdd = pr.make_precondense_batch(L, grids, 1)
    D = Records(L, "dir")
    rng = np.random.default_rng(7)
    d_in = D.zeros(1, len(grids))
    for f in ("dx", "du", "dlmdgmm", "dxi"):
        D.f(d_in, f)[...] = 0.3 * rng.uniform(-1, 1, D.f(d_in, f).shape)
    kkt_out, cdd_out, d_out = kkt.copy(), cdd.copy(), d_in.copy()
    for i, g in enumerate(grids):
        if g.type == GRID_TERMINAL:
            continue
        ref.condense_stage(L, g, kkt_out[0, i], cdd_out[0, i])
    cdd_cond = cdd_out.copy()
    for i, g in enumerate(grids):
        if g.type == GRID_TERMINAL:
            continue
        ref.expand_stage(L, g, cdd_out[0, i], d_out[0, i], d_out[0, i + 1])
    np.savez_compressed(os.path.join(HERE, name), kkt=kkt, cdd=cdd, dir_in=d_in, kkt_out=kkt_out, cdd_condensed=cdd_cond,
                        cdd_out=cdd_out, dir_out=d_out, **grid_table(grids))


def main():
    ref.build()
    riccati_fixture("ref_anymal_trot_n8_riccati.npz", *anymal_trot_short(), mode="factory")
    riccati_fixture("ref_anymal_jump_sto_n8_riccati.npz", *anymal_jump_sto_short(), mode="dynamics")
    condense_fixture("ref_anymal_trot_n8_condense.npz", *anymal_trot_short())
    dims2, grids2, meta = pr.config_iiwa14()
    L2 = orc.layout(dims2)
    n = len(grids2)
    kkt2 = Records(L2, "kkt").zeros(1, n)
    pr.fill_unconstr_instance(L2, n, kkt2[0], np.random.default_rng(pr.BASE_SEED))
    dx02 = pr.make_dx0(L2, 1)
    ric2, d2 = Records(L2, "ric").zeros(1, n), Records(L2, "dir").zeros(1, n)
    Records(L2, "dir").f(d2[0, 0], "dx")[...] = dx02[0]
    ref.unconstr_sweep(L2, n, meta["dt"], kkt2[0].copy(), ric2[0], d2[0])
    np.savez_compressed(os.path.join(HERE, "ref_iiwa14_unconstr_riccati.npz"), kkt=kkt2, dx0=dx02, ric=ric2, dir=d2, dt=meta["dt"])
    for f in sorted(os.listdir(HERE)):
        if f.startswith("ref_"):
            print(f, os.path.getsize(os.path.join(HERE, f)))





def unconstr_solver_fixture(name, with_limits):
    """One UnconstrOCPSolver::updateSolution of the iiwa14 OCP (BASELINE configs[0]) run by the REFERENCE'S OWN sources --
    stages, ConfigurationSpaceCost, joint-limit components, forward-Euler state equation, UnconstrDynamics, Riccati recursion,
    step sizes, update (oracle/ref_shim/ref_unconstr_solver_capi.cpp) -- with the inverse dynamics and its partial
    derivatives of every grid point injected from this repository's CPU rigid-body restatement (Richardson-extrapolated
    central differences: ~1e-11).  Inputs and outputs for the GPU test, where /root/reference does not exist."""
    from robotoc_amd import robot_model as rm
    m = rm.load_named("iiwa14")
    nv, N, dt = m.nv, 20, 0.05
    rng = np.random.default_rng(4242 + int(with_limits))
    cost = np.stack([rng.uniform(-0.8, 0.8, nv), np.zeros(nv), np.zeros(nv), np.full(nv, 10.0), np.full(nv, 0.1), np.full(nv, 0.01),
                     np.full(nv, 0.001), np.full(nv, 10.0), np.full(nv, 0.1)])
    x0 = np.concatenate([rng.uniform(-0.5, 0.5, nv), np.zeros(nv)])
    sol = np.zeros((N + 1, 7 * nv))
    sol[:, :nv] = x0[:nv] + 0.1 * rng.uniform(-1, 1, (N + 1, nv))
    sol[:, nv:2 * nv] = 0.3 * rng.uniform(-1, 1, (N + 1, nv))
    sol[:, 2 * nv:3 * nv] = rng.uniform(-1, 1, (N + 1, nv))
    sol[:, 3 * nv:4 * nv] = 5.0 * rng.uniform(-1, 1, (N + 1, nv))
    sol[:, 4 * nv:] = 0.5 * rng.uniform(-1, 1, (N + 1, 3 * nv))
    limits = np.stack([np.full(nv, -1.0), np.full(nv, 1.0), np.full(nv, 1.5), np.full(nv, 40.0)]) if with_limits else None
    rnea = np.zeros((N, nv + 3 * nv * nv))
    z = np.zeros(0)
    for i in range(N):
        q, v, a = sol[i, :nv], sol[i, nv:2 * nv], sol[i, 2 * nv:3 * nv]
        rnea[i, :nv] = orc.rbd_eval(m, 0, q, v, a, z, np.zeros(nv), 0, z)[:nv]
        h = 2.0e-3
        J1, J2 = orc.rbd_linearize_fd(m, 0, q, v, a, z, np.zeros(nv), 0, z, eps=h), orc.rbd_linearize_fd(m, 0, q, v, a, z, np.zeros(nv), 0, z, eps=h / 2)
        for k in range(3):   # dID/dq, dID/dv, dID/da: Richardson step on the central differences
            J = (4.0 * np.asarray(J2[k])[:nv] - np.asarray(J1[k])[:nv]) / 3.0
            rnea[i, nv + k * nv * nv:nv + (k + 1) * nv * nv] = J.T.reshape(-1)   # column-major
    sol_in = sol.copy()
    kkt, err, primal, dual, con = ref.unconstr_update_solution(nv, N, dt, cost, limits, 1.0e-3, 0.995, x0, sol, rnea)
    np.savez_compressed(os.path.join(HERE, name), cost=cost, x0=x0, sol_in=sol_in, sol_out=sol, rnea=rnea, kkt=kkt,
                        limits=limits if with_limits else np.zeros(0), scalars=np.array([err, primal, dual, dt, 1.0e-3, 0.995]), con=con)
    print(name, "KKT error %.6e, steps %.4f / %.4f" % (err, primal, dual))


def unconstr_line_search_fixture(name, with_limits, backtrack):
    """One UnconstrOCPSolver::updateSolution of the iiwa14 OCP WITH SolverOptions::enable_line_search
    (unconstr_ocp_solver.cpp:96-118), run by the REFERENCE'S OWN UnconstrDirectMultipleShooting, UnconstrRiccatiRecursion and
    UnconstrLineSearch (src/line_search/unconstr_line_search.cpp; oracle/ref_shim/ref_unconstr_ls_capi.cpp).  The inverse
    dynamics of the iterate (with its partial derivatives) and of every trial iterate the filter loop visits is injected from this
    repository's CPU rigid-body restatement.  backtrack = False: a random inconsistent iterate (the full step closes most of the
    violation: accepted at the first trial).  backtrack = True: the arm at rest in a bent pose that is also the cost's reference,
    zero torques, torques weighted a thousand times the configuration -- the Newton step swings the arm by more than a radian to where
    the LINEARISED gravity torque vanishes, the cost rises, the real violation does not fall by the filter's margin: the reference
    rejects several trials before it accepts."""
    from robotoc_amd import robot_model as rm
    m = rm.load_named("iiwa14")
    nv, N, dt = m.nv, 20, 0.05
    rng = np.random.default_rng(9090 + int(with_limits) + 10 * int(backtrack))
    sol = np.zeros((N + 1, 7 * nv))
    if backtrack:
        q = 1.5 * rng.uniform(-1, 1, nv)
        cost = np.stack([q, np.zeros(nv), np.zeros(nv), np.full(nv, 0.01), np.full(nv, 0.1), np.full(nv, 0.01), np.full(nv, 10.0),
                         np.full(nv, 0.01), np.full(nv, 0.1)])
        x0 = np.concatenate([q, np.zeros(nv)])
        sol[:, :nv] = q
        limits = np.stack([np.full(nv, -4.0), np.full(nv, 4.0), np.full(nv, 30.0), np.full(nv, 500.0)]) if with_limits else None
    else:
        cost = np.stack([rng.uniform(-0.8, 0.8, nv), np.zeros(nv), np.zeros(nv), np.full(nv, 10.0), np.full(nv, 0.1), np.full(nv, 0.01),
                         np.full(nv, 0.001), np.full(nv, 10.0), np.full(nv, 0.1)])
        x0 = np.concatenate([rng.uniform(-0.5, 0.5, nv), np.zeros(nv)])
        sol[:, :nv] = x0[:nv] + 0.1 * rng.uniform(-1, 1, (N + 1, nv))
        sol[:, nv:2 * nv] = 0.3 * rng.uniform(-1, 1, (N + 1, nv))
        sol[:, 2 * nv:3 * nv] = rng.uniform(-1, 1, (N + 1, nv))
        sol[:, 3 * nv:4 * nv] = 5.0 * rng.uniform(-1, 1, (N + 1, nv))
        sol[:, 4 * nv:] = 0.5 * rng.uniform(-1, 1, (N + 1, 3 * nv))
        limits = np.stack([np.full(nv, -1.0), np.full(nv, 1.0), np.full(nv, 1.5), np.full(nv, 40.0)]) if with_limits else None
    rnea = np.zeros((N, nv + 3 * nv * nv))
    z = np.zeros(0)

    def inverse_dynamics(q, v, a):
        return orc.rbd_eval(m, 0, q, v, a, z, np.zeros(nv), 0, z)[:nv]
    for i in range(N):
        q, v, a = sol[i, :nv], sol[i, nv:2 * nv], sol[i, 2 * nv:3 * nv]
        rnea[i, :nv] = inverse_dynamics(q, v, a)
        h = 2.0e-3
        J1, J2 = orc.rbd_linearize_fd(m, 0, q, v, a, z, np.zeros(nv), 0, z, eps=h), orc.rbd_linearize_fd(m, 0, q, v, a, z, np.zeros(nv), 0, z, eps=h / 2)
        for k in range(3):
            J = (4.0 * np.asarray(J2[k])[:nv] - np.asarray(J1[k])[:nv]) / 3.0
            rnea[i, nv + k * nv * nv:nv + (k + 1) * nv * nv] = J.T.reshape(-1)
    sol_in = sol.copy()
    settings = (0.75, 0.05, 0.005, 0.005)   # LineSearchSettings' defaults (line_search_settings.hpp)
    r = ref.unconstr_line_search_iteration(nv, N, dt, cost, limits, 1.0e-3, 0.995, x0, sol, rnea, inverse_dynamics, settings)
    np.savez_compressed(os.path.join(HERE, name), cost=cost, x0=x0, sol_in=sol_in, sol_out=sol, limits=limits if with_limits else np.zeros(0),
                        direction=r["direction"], dslack=r["dslack"], con=r["con"], settings=np.array(settings),
                        scalars=np.array([r["kkt_error"], r["max_primal"], r["max_dual"], dt, 1.0e-3, 0.995, r["step"], r["trials"]]),
                        eval=r["eval"])
    print(name, "KKT error %.6e, max steps %.4f / %.4f, accepted %.6f after %d trials; cost %.4e barrier %.4e violation %.4e"
          % (r["kkt_error"], r["max_primal"], r["max_dual"], r["step"], r["trials"], *r["eval"]))


def _richardson(fun, n, h=2.0e-3):
    """Jacobian of fun over an n-dimensional perturbation: central differences at h and h / 2, one Richardson step (O(h^4))"""
    def central(step):
        cols = []
        for k in range(n):
            e = np.zeros(n)
            e[k] = step
            cols.append((np.asarray(fun(e)) - np.asarray(fun(-e))) / (2.0 * step))
        return np.stack(cols, axis=-1)
    return (4.0 * central(h / 2) - central(h)) / 3.0


def contact_stage_fixture(name="ref_anymal_contact_stage.npz"):
    """IntermediateStage::evalKKT of ANYmal on four feet (ConfigurationSpaceCost, six joint-limit components, FrictionCone), run
    by the REFERENCE'S OWN stage / cost / constraints / state-equation / contact-dynamics sources
    (oracle/ref_shim/ref_contact_stage_capi.cpp); every Pinocchio quantity injected from this repository's CPU restatement
    (Richardson-extrapolated differences for the Jacobians).  Grid point 2 of a five-point horizon: inputs for the device,
    the reference's condensed KKT blocks as expected outputs."""
    import ctypes as C
    from robotoc_amd import robot_model as rm
    from robotoc_amd.grid import ANYMAL_Q_STANDING
    m = rm.load_named("anymal")
    nv, nq, nu, nc, n, i0, dt = m.nv, m.nq, 12, 4, 5, 2, 0.02
    rng = np.random.default_rng(777)
    qs = np.array(ANYMAL_Q_STANDING, dtype=float)
    feet = np.array([orc.rbd_contact_position(m, qs, c) for c in range(nc)]) + 0.005 * rng.uniform(-1, 1, (nc, 3))
    q = np.zeros((n, nq))
    for i in range(n):
        q[i] = qs
        q[i, :7] = orc.se3_integrate(qs[:7], 0.05 * rng.uniform(-1, 1, 6))
        q[i, 7:] += 0.1 * rng.uniform(-1, 1, 12)
    v, a, u = 0.5 * rng.uniform(-1, 1, (n, nv)), rng.uniform(-1, 1, (n, nv)), 10.0 * rng.uniform(-1, 1, (n, nu))
    f = 10.0 * rng.uniform(-1, 1, (n, nc, 3))
    f[:, :, 2] = rng.uniform(40, 90, (n, nc))
    lmd, gmm, beta = (0.5 * rng.uniform(-1, 1, (n, nv)) for _ in range(3))
    mus, nup = 0.5 * rng.uniform(-1, 1, (n, nc, 3)), 0.5 * rng.uniform(-1, 1, (n, 6))
    mu = np.array([0.7, 0.6, 0.8, 0.5])
    q_ref = qs.copy()
    q_ref[:7] = orc.se3_integrate(qs[:7], np.array([0.03, 0.0, -0.02, 0.0, 0.05, 0.0]))
    M = nv + 1
    cost = np.zeros((12, M))
    for k, val in ((0, q_ref), (3, np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])), (4, np.full(nv, 1.0)), (5, np.full(nv, 1e-3)),
                   (6, np.full(nu, 1e-3)), (7, np.full(nv, 10.0)), (8, np.full(nv, 1.0))):
        cost[k, :len(val)] = val
    limits = np.stack([np.full(nu, -1.6), np.full(nu, 1.6), np.full(nu, 3.0), np.full(nu, 40.0)])
    nrow = 6 * nu + 5 * nc
    slack, dual = rng.uniform(0.1, 2.0, nrow), rng.uniform(0.01, 0.5, nrow)
    barrier, tau = 1.0e-3, 0.995
    active = 0b1111

    def sub(qf, q0):
        return np.concatenate([orc.se3_difference(q0[:7], qf[:7]), qf[7:] - q0[7:]])
    qi, qn, qp = q[i0], q[i0 + 1], q[i0 - 1]
    plus = lambda qq, e: orc.rbd_integrate(m, qq, e)
    L = ref.lib()
    dp = C.POINTER(C.c_double)
    ptr = lambda x: np.ascontiguousarray(x, dtype=np.float64).ctypes.data_as(dp)
    L.ref_stage_begin(nv, nu, nc)

    def inject(key, arr):
        arr = np.asfortranarray(np.atleast_2d(np.asarray(arr, dtype=np.float64).T).T if np.ndim(arr) == 1 else arr)
        if arr.ndim == 1 or arr.shape[1] == 0:
            arr = arr.reshape(-1, 1)
        keep = np.asfortranarray(arr)
        assert L.ref_stage_inject(key.encode(), keep.ctypes.data_as(dp), keep.shape[0], keep.shape[1]) == 0
    # the order IntermediateStage::evalKKT asks for them (intermediate_stage.cpp:94-148)
    inject("subtractConfiguration", sub(qi, q_ref).reshape(-1, 1))                                  # cost (:277 / hpp:193)
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), q_ref), nv))       # cost (hpp:214)
    inject("subtractConfiguration", sub(qi, qn).reshape(-1, 1))                                     # state_equation.cpp:16
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), qn), nv))          # :39
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qp, plus(qi, e)), nv))          # :41
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qi, plus(qn, e)), nv))          # :78
    fstack = f[i0].reshape(-1)
    val = orc.rbd_eval(m, 0, qi, v[i0], a[i0], fstack, np.zeros(nu), active, feet.reshape(-1))
    h = 2.0e-3
    J1 = orc.rbd_linearize_fd(m, 0, qi, v[i0], a[i0], fstack, np.zeros(nu), active, feet.reshape(-1), eps=h)
    J2 = orc.rbd_linearize_fd(m, 0, qi, v[i0], a[i0], fstack, np.zeros(nu), active, feet.reshape(-1), eps=h / 2)
    Jr = [(4.0 * np.asarray(J2[k]) - np.asarray(J1[k])) / 3.0 for k in range(3)]   # [dID; dC] / d(q, v, a): (nv + 12) x nv
    assert L.ref_stage_inverse_dynamics(ptr(val[:nv]), ptr(np.asfortranarray(Jr[0][:nv]).T.copy()), ptr(np.asfortranarray(Jr[1][:nv]).T.copy()),
                                        ptr(np.asfortranarray(Jr[2][:nv]).T.copy())) == 0
    inject("baumgarteResidual", val[nv:].reshape(-1, 1))
    for k, key in enumerate(("baumgarte_dq", "baumgarte_dv", "baumgarte_da")):
        inject(key, Jr[k][nv:])
    frames = []
    for c in range(nc):
        R = orc.rbd_contact_placement(m, qi, c)[0]
        dR = _richardson(lambda e: orc.rbd_contact_placement(m, plus(qi, e), c)[0].reshape(-1), nv)
        Jl = np.zeros((6, nv))
        for jj in range(nv):
            W = dR[:, jj].reshape(3, 3) @ R.T
            Jl[3:, jj] = R.T @ np.array([W[2, 1], W[0, 2], W[1, 0]])   # LOCAL-frame angular Jacobian column
        assert L.ref_stage_frame(c, ptr(R), ptr(Jl.T.copy())) == 0
        frames.append((R, Jl))
    sol = np.concatenate([qi, v[i0], a[i0], u[i0], f[i0].reshape(-1), lmd[i0], gmm[i0], beta[i0], mus[i0].reshape(-1), nup[i0]])
    sol_next = np.concatenate([qn, v[i0 + 1], lmd[i0 + 1], gmm[i0 + 1]])
    nx = 2 * nv
    out = np.zeros(nx * nx + nx * nu + nu * nu + nx * nx + nv * nu + nx + nu + nx + nx + nu + nx + 4)
    L.ref_contact_stage_eval_kkt.argtypes = [C.c_uint, dp, dp, C.c_double, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp]
    rc = L.ref_contact_stage_eval_kkt(active, ptr(feet), ptr(mu), dt, i0, n - 1, ptr(cost), ptr(limits), barrier, tau, ptr(qp), ptr(sol),
                                      ptr(sol_next), ptr(slack), ptr(dual), out.ctypes.data_as(dp))
    assert rc == 0, rc
    np.savez_compressed(os.path.join(HERE, name), q=q, v=v, a=a, u=u, f=f, lmd=lmd, gmm=gmm, beta=beta, mu_stack=mus, nu_passive=nup,
                        feet=feet, mu=mu, cost=cost, limits=limits, slack=slack, dual=dual, out=out,
                        scalars=np.array([dt, barrier, tau, i0, active]))
    print(name, "stage KKT error %.6e, h %.6e" % (out[-1], out[-2]))


def impact_and_terminal_stage_fixture(name="ref_anymal_impact_terminal_stage.npz"):
    """ImpactStage::evalKKT and TerminalStage::evalKKT (src/ocp/impact_stage.cpp:78-120, terminal_stage.cpp:70-100) of ANYmal
    touching down with two feet, by the REFERENCE'S OWN sources with injected rigid-body quantities
    (oracle/ref_shim/ref_contact_stage_capi.cpp: ref_contact_stage_eval_kkt2).  The horizon: two feet, touch-down of the other two,
    four feet; the impact grid and the terminal grid are the ones compared."""
    import ctypes as C
    from robotoc_amd import robot_model as rm
    from robotoc_amd.grid import ANYMAL_Q_STANDING, ContactSequence, Event, contact_masks
    from robotoc_amd.types import GRID_IMPACT
    m = rm.load_named("anymal")
    nv, nq, nu, nc = m.nv, m.nq, 12, 4
    grids = discretize(8, 0.16, 0.0, ContactSequence([6, 12], [Event("impact", 0.07, impact_dimf=6)]))
    n = len(grids)
    masks = contact_masks(grids, [0b1001, 0b1111], [0b0110])
    i_imp = [i for i, g in enumerate(grids) if g.type == GRID_IMPACT][0]
    rng = np.random.default_rng(4321)
    qs = np.array(ANYMAL_Q_STANDING, dtype=float)
    feet = np.array([orc.rbd_contact_position(m, qs, c) for c in range(nc)]) + 0.005 * rng.uniform(-1, 1, (nc, 3))
    q = np.zeros((n, nq))
    for i in range(n):
        q[i] = qs
        q[i, :7] = orc.se3_integrate(qs[:7], 0.05 * rng.uniform(-1, 1, 6))
        q[i, 7:] += 0.1 * rng.uniform(-1, 1, 12)
    v, a, u = 0.5 * rng.uniform(-1, 1, (n, nv)), rng.uniform(-1, 1, (n, nv)), 10.0 * rng.uniform(-1, 1, (n, nu))
    f = 10.0 * rng.uniform(-1, 1, (n, nc, 3))
    f[:, :, 2] = rng.uniform(40, 90, (n, nc))
    lmd, gmm, beta = (0.5 * rng.uniform(-1, 1, (n, nv)) for _ in range(3))
    mus = 0.5 * rng.uniform(-1, 1, (n, nc, 3))
    mu = np.array([0.7, 0.6, 0.8, 0.5])
    q_ref = qs.copy()
    q_ref[:7] = orc.se3_integrate(qs[:7], np.array([0.03, 0.0, -0.02, 0.0, 0.05, 0.0]))
    M = nv + 1
    cost = np.zeros((12, M))
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])
    for k, val in ((0, q_ref), (3, wq), (4, np.full(nv, 1.0)), (5, np.full(nv, 1e-3)), (6, np.full(nu, 1e-3)), (7, 10.0 * wq), (8, np.full(nv, 2.0)),
                   (9, 3.0 * wq), (10, np.full(nv, 0.5)), (11, np.full(nv, 0.05))):
        cost[k, :len(val)] = val
    limits = np.stack([np.full(nu, -1.6), np.full(nu, 1.6), np.full(nu, 3.0), np.full(nu, 40.0)])
    barrier, tau = 1.0e-3, 0.995
    L = ref.lib()
    dp = C.POINTER(C.c_double)
    ptr = lambda x: np.ascontiguousarray(x, dtype=np.float64).ctypes.data_as(dp)
    L.ref_contact_stage_eval_kkt2.argtypes = [C.c_int, C.c_uint, C.c_uint, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp]
    plus = lambda qq, e: orc.rbd_integrate(m, qq, e)

    def sub(qf, q0):
        return np.concatenate([orc.se3_difference(q0[:7], qf[:7]), qf[7:] - q0[7:]])

    def inject(key, arr):
        keep = np.asfortranarray(np.asarray(arr, dtype=np.float64).reshape(len(arr), -1))
        assert L.ref_stage_inject(key.encode(), keep.ctypes.data_as(dp), keep.shape[0], keep.shape[1]) == 0
    nx = 2 * nv
    # ---- the impact grid ----
    L.ref_stage_begin(nv, nu, nc)
    qi, qn, qp = q[i_imp], q[i_imp + 1], q[i_imp - 1]
    inject("subtractConfiguration", sub(qi, q_ref))
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), q_ref), nv))
    inject("subtractConfiguration", sub(qi, qn))
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), qn), nv))
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qp, plus(qi, e)), nv))
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qi, plus(qn, e)), nv))
    imp_mask = int(masks[i_imp])
    imp = [c for c in range(nc) if (imp_mask >> c) & 1]
    fstack = np.concatenate([f[i_imp, c] for c in imp])
    val = orc.rbd_eval(m, 1, qi, v[i_imp], a[i_imp], fstack, np.zeros(nu), imp_mask, feet.reshape(-1))
    h = 2.0e-3
    J1 = orc.rbd_linearize_fd(m, 1, qi, v[i_imp], a[i_imp], fstack, np.zeros(nu), imp_mask, feet.reshape(-1), eps=h)
    J2 = orc.rbd_linearize_fd(m, 1, qi, v[i_imp], a[i_imp], fstack, np.zeros(nu), imp_mask, feet.reshape(-1), eps=h / 2)
    Jr = [(4.0 * np.asarray(J2[k]) - np.asarray(J1[k])) / 3.0 for k in range(3)]
    cm = lambda X: np.asfortranarray(X).T.copy()
    assert L.ref_stage_inverse_dynamics(ptr(val[:nv]), ptr(cm(Jr[0][:nv])), ptr(cm(Jr[1][:nv])), ptr(cm(Jr[2][:nv]))) == 0
    inject("impactVelocityResidual", val[nv:])
    inject("impactVelocity_dq", Jr[0][nv:])
    inject("impactVelocity_dv", Jr[2][nv:])      # the rows see v + dv: d/dv = d/d(dv)
    fi = np.zeros((nc, 3))
    mi = np.zeros((nc, 3))
    for c in imp:
        fi[c], mi[c] = f[i_imp, c], mus[i_imp, c]
    sol = np.concatenate([qi, v[i_imp], a[i_imp], fi.reshape(-1), lmd[i_imp], gmm[i_imp], beta[i_imp], mi.reshape(-1)])
    sol_next = np.concatenate([qn, v[i_imp + 1], lmd[i_imp + 1], gmm[i_imp + 1]])
    out_imp = np.zeros(2 * nx * nx + 2 * nx)
    rc = L.ref_contact_stage_eval_kkt2(1, 0b1001, imp_mask, ptr(feet), ptr(mu), ptr(cost), ptr(limits), barrier, tau, ptr(qp), ptr(sol), ptr(sol_next),
                                       out_imp.ctypes.data_as(dp))
    assert rc == 0, rc
    # ---- the terminal grid ----
    L.ref_stage_begin(nv, nu, nc)
    qt, qpt = q[n - 1], q[n - 2]
    inject("subtractConfiguration", sub(qt, q_ref))
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qt, e), q_ref), nv))
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qpt, plus(qt, e)), nv))
    sol_t = np.concatenate([qt, v[n - 1], lmd[n - 1], gmm[n - 1]])
    out_term = np.zeros(nx * nx + nx)
    rc = L.ref_contact_stage_eval_kkt2(2, 0b1001, imp_mask, ptr(feet), ptr(mu), ptr(cost), ptr(limits), barrier, tau, ptr(qpt), ptr(sol_t), ptr(sol_t),
                                       out_term.ctypes.data_as(dp))
    assert rc == 0, rc
    # ---- the grid point two ahead of the touch-down: switching constraint (no inequality rows) ----
    L.ref_stage_begin(nv, nu, nc)
    L.ref_contact_stage_eval_kkt3.argtypes = [C.c_uint, C.c_uint, dp, dp, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, dp]
    i_sw = i_imp - 2
    assert grids[i_sw].switching_constraint and grids[i_sw].dims == 6
    qi, qn, qp = q[i_sw], q[i_sw + 1], q[i_sw - 1]
    dt1, dt2 = grids[i_sw].dt, grids[i_sw + 1].dt
    act_mask = int(masks[i_sw])
    act = [c for c in range(nc) if (act_mask >> c) & 1]
    xi = 0.5 * rng.uniform(-1, 1, 6)
    inject("subtractConfiguration", sub(qi, q_ref))
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), q_ref), nv))
    inject("subtractConfiguration", sub(qi, qn))
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), qn), nv))
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qp, plus(qi, e)), nv))
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qi, plus(qn, e)), nv))
    fstack = np.concatenate([f[i_sw, c] for c in act])
    val = orc.rbd_eval(m, 0, qi, v[i_sw], a[i_sw], fstack, np.zeros(nu), act_mask, feet.reshape(-1))
    J1 = orc.rbd_linearize_fd(m, 0, qi, v[i_sw], a[i_sw], fstack, np.zeros(nu), act_mask, feet.reshape(-1), eps=h)
    J2 = orc.rbd_linearize_fd(m, 0, qi, v[i_sw], a[i_sw], fstack, np.zeros(nu), act_mask, feet.reshape(-1), eps=h / 2)
    Jr = [(4.0 * np.asarray(J2[k]) - np.asarray(J1[k])) / 3.0 for k in range(3)]
    assert L.ref_stage_inverse_dynamics(ptr(val[:nv]), ptr(cm(Jr[0][:nv])), ptr(cm(Jr[1][:nv])), ptr(cm(Jr[2][:nv]))) == 0
    inject("baumgarteResidual", val[nv:])
    for k, key in enumerate(("baumgarte_dq", "baumgarte_dv", "baumgarte_da")):
        inject(key, Jr[k][nv:])
    dq_sw = (dt1 + dt2) * v[i_sw] + dt1 * dt2 * a[i_sw]
    q_plus = plus(qi, dq_sw)
    P_at = lambda qq: np.concatenate([orc.rbd_contact_position(m, qq, c) - feet[c] for c in imp])
    inject("integrateConfiguration", q_plus)
    inject("contactPositionResidual", P_at(q_plus))
    inject("contactPositionDerivative", _richardson(lambda e: P_at(plus(q_plus, e)), nv))
    inject("dIntegrate_dq", _richardson(lambda e: sub(plus(plus(qi, e), dq_sw), q_plus), nv))
    inject("dIntegrate_dv", _richardson(lambda e: sub(plus(qi, dq_sw + e), q_plus), nv))
    fs = np.zeros((nc, 3))
    ms = np.zeros((nc, 3))
    for c in act:
        fs[c], ms[c] = f[i_sw, c], mus[i_sw, c]
    nup = 0.5 * rng.uniform(-1, 1, 6)
    sol = np.concatenate([qi, v[i_sw], a[i_sw], u[i_sw], fs.reshape(-1), lmd[i_sw], gmm[i_sw], beta[i_sw], ms.reshape(-1), nup, xi])
    sol_next = np.concatenate([qn, v[i_sw + 1], lmd[i_sw + 1], gmm[i_sw + 1]])
    ns = 6
    out_sw = np.zeros(nx * nx + nx * nu + nu * nu + nx * nx + nv * nu + nx + nu + nx + ns * nx + ns * nu + ns + ns + nx + nu + 2)
    rc = L.ref_contact_stage_eval_kkt3(act_mask, imp_mask, ptr(feet), ptr(mu), dt1, dt2, grids[i_sw].time_stage, grids[i_sw].num_grids_in_phase,
                                       ptr(cost), ptr(qp), ptr(sol), ptr(sol_next), out_sw.ctypes.data_as(dp))
    assert rc == 0, rc
    np.savez_compressed(os.path.join(HERE, name), q=q, v=v, a=a, u=u, f=f, lmd=lmd, gmm=gmm, beta=beta, mu_stack=mus, feet=feet, mu=mu, cost=cost,
                        limits=limits, out_impact=out_imp, out_terminal=out_term, out_switching=out_sw, xi=xi, nu_passive=nup,
                        scalars=np.array([barrier, tau, i_imp]), **grid_table(grids))
    print(name, "impact grid", i_imp, "of", n)


def icub_surface_stage_fixture(name="ref_icub_surface_stage.npz"):
    """IntermediateStage::evalKKT of iCub (nv = 35) on its two soles -- SURFACE contacts: six contact rows each with the Log6
    placement error, ContactWrenchCone (17 rows per sole), six joint-limit components, ConfigurationSpaceCost -- by the REFERENCE'S
    OWN sources with injected rigid-body quantities (ref_contact_stage_eval_kkt_surface).  Grid point 2 of a five-point horizon."""
    import ctypes as C
    from robotoc_amd import robot_model as rm
    m = rm.load_named("icub")
    nv, nq, nu, nc, n, i0, dt = m.nv, m.nq, m.nv - 6, 2, 5, 2, 0.02
    rng = np.random.default_rng(99)
    qs = np.array([0, 0, 0.592, 0, 0, 1, 0, 0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726, 0.20944, 0.08727, 0, -0.1745, -0.0279, -0.08726,
                   0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0, 0, 0.35, 0.5, 0.5, 0, 0, 0])
    place = [orc.rbd_contact_placement(m, qs, c) for c in range(nc)]
    pos = np.array([p for _, p in place]) + 0.003 * rng.uniform(-1, 1, (nc, 3))
    rot = np.array([R @ orc.rbd_exp6(np.concatenate([np.zeros(3), 0.02 * rng.uniform(-1, 1, 3)]))[0] for R, _ in place])
    q = np.zeros((n, nq))
    for i in range(n):
        q[i] = qs
        q[i, :7] = orc.se3_integrate(qs[:7], 0.03 * rng.uniform(-1, 1, 6))
        q[i, 7:] += 0.05 * rng.uniform(-1, 1, nu)
    v, a, u = 0.3 * rng.uniform(-1, 1, (n, nv)), rng.uniform(-1, 1, (n, nv)), 10.0 * rng.uniform(-1, 1, (n, nu))
    f = 5.0 * rng.uniform(-1, 1, (n, nc, 6))
    f[:, :, 2] = rng.uniform(100, 200, (n, nc))
    lmd, gmm, beta = (0.5 * rng.uniform(-1, 1, (n, nv)) for _ in range(3))
    mus, nup = 0.5 * rng.uniform(-1, 1, (n, nc, 6)), 0.5 * rng.uniform(-1, 1, (n, 6))
    mu = np.array([0.6, 0.8])
    X, Y = 0.1, 0.05
    q_ref = qs.copy()
    q_ref[2] -= 0.03
    M = nv + 1
    cost = np.zeros((12, M))
    for k, val in ((0, q_ref), (3, np.concatenate([np.full(6, 10.0), np.full(nu, 0.1)])), (4, np.full(nv, 0.1)), (5, np.full(nv, 1e-3)),
                   (6, np.full(nu, 1e-4)), (7, np.full(nv, 10.0)), (8, np.full(nv, 0.1))):
        cost[k, :len(val)] = val
    limits = np.stack([np.full(nu, -2.5), np.full(nu, 2.5), np.full(nu, 5.0), np.full(nu, 60.0)])
    nrow = 6 * nu + 17 * nc
    slack, dual = rng.uniform(0.1, 2.0, nrow), rng.uniform(0.01, 0.5, nrow)
    barrier, tau, active = 1.0e-3, 0.995, 0b11
    qi, qn, qp = q[i0], q[i0 + 1], q[i0 - 1]
    plus = lambda qq, e: orc.rbd_integrate(m, qq, e)

    def sub(qf, q0):
        return np.concatenate([orc.se3_difference(q0[:7], qf[:7]), qf[7:] - q0[7:]])
    L = ref.lib()
    dp = C.POINTER(C.c_double)
    ptr = lambda x: np.ascontiguousarray(x, dtype=np.float64).ctypes.data_as(dp)
    L.ref_stage_begin_surface(nv, nu, nc)

    def inject(key, arr):
        keep = np.asfortranarray(np.asarray(arr, dtype=np.float64).reshape(len(arr), -1))
        assert L.ref_stage_inject(key.encode(), keep.ctypes.data_as(dp), keep.shape[0], keep.shape[1]) == 0
    inject("subtractConfiguration", sub(qi, q_ref))
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), q_ref), nv))
    inject("subtractConfiguration", sub(qi, qn))
    inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), qn), nv))
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qp, plus(qi, e)), nv))
    inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qi, plus(qn, e)), nv))
    fstack = f[i0].reshape(-1)
    val = orc.rbd_eval(m, 0, qi, v[i0], a[i0], fstack, np.zeros(nu), active, pos.reshape(-1), rot.reshape(nc, 9))
    h = 2.0e-3
    J1 = orc.rbd_linearize_fd(m, 0, qi, v[i0], a[i0], fstack, np.zeros(nu), active, pos.reshape(-1), eps=h, rref=rot.reshape(nc, 9))
    J2 = orc.rbd_linearize_fd(m, 0, qi, v[i0], a[i0], fstack, np.zeros(nu), active, pos.reshape(-1), eps=h / 2, rref=rot.reshape(nc, 9))
    Jr = [(4.0 * np.asarray(J2[k]) - np.asarray(J1[k])) / 3.0 for k in range(3)]
    cm = lambda Xm: np.asfortranarray(Xm).T.copy()
    assert L.ref_stage_inverse_dynamics(ptr(val[:nv]), ptr(cm(Jr[0][:nv])), ptr(cm(Jr[1][:nv])), ptr(cm(Jr[2][:nv]))) == 0
    inject("baumgarteResidual", val[nv:])
    for k, key in enumerate(("baumgarte_dq", "baumgarte_dv", "baumgarte_da")):
        inject(key, Jr[k][nv:])
    # the wrench cone reads no kinematics; frames only for completeness
    for c in range(nc):
        assert L.ref_stage_frame(c, ptr(orc.rbd_contact_placement(m, qi, c)[0]), ptr(np.zeros((nv, 6)))) == 0
    sol = np.concatenate([qi, v[i0], a[i0], u[i0], f[i0].reshape(-1), lmd[i0], gmm[i0], beta[i0], mus[i0].reshape(-1), nup[i0]])
    sol_next = np.concatenate([qn, v[i0 + 1], lmd[i0 + 1], gmm[i0 + 1]])
    nx = 2 * nv
    out = np.zeros(nx * nx + nx * nu + nu * nu + nx * nx + nv * nu + nx + nu + nx + nx + nu + nx + 4)
    L.ref_contact_stage_eval_kkt_surface.argtypes = [C.c_int, C.c_double, C.c_double, dp, C.c_uint, dp, dp, C.c_double, C.c_int, C.c_int, dp, dp,
                                                     C.c_double, C.c_double, dp, dp, dp, dp, dp, dp]
    rc = L.ref_contact_stage_eval_kkt_surface(2, X, Y, ptr(rot.reshape(-1)), active, ptr(pos), ptr(mu), dt, i0, n - 1, ptr(cost), ptr(limits), barrier,
                                              tau, ptr(qp), ptr(sol), ptr(sol_next), ptr(slack), ptr(dual), out.ctypes.data_as(dp))
    assert rc == 0, rc
    np.savez_compressed(os.path.join(HERE, name), q=q, v=v, a=a, u=u, f=f, lmd=lmd, gmm=gmm, beta=beta, mu_stack=mus, nu_passive=nup, pos=pos, rot=rot,
                        mu=mu, cost=cost, limits=limits, slack=slack, dual=dual, out=out, scalars=np.array([dt, barrier, tau, i0, active, X, Y]))
    print(name, "stage KKT error %.6e" % out[-1])


def ocp_solver_iteration_fixture(name="ref_anymal_ocp_solver_iteration.npz", sto=False, line_search=False, armijo=0.001, robot="anymal"):
    """ONE OCPSolver::updateSolution (src/solver/ocp_solver.cpp:111-145) of ANYmal over a short trot -- lifts, touch-downs with
    switching constraints, ConfigurationSpaceCost, six joint-limit components, FrictionCone -- run by the REFERENCE'S OWN
    DirectMultipleShooting, stages, ContactSequence, cost, constraints, dynamics and RiccatiRecursion sources
    (oracle/ref_shim/ref_ocp_capi.cpp), every Pinocchio quantity of every grid point injected from this repository's CPU
    restatement.  The fixture: the iterate, and the reference's next iterate, slacks / duals, step sizes and KKT error.
    sto=True: BASELINE configs[2] at its size -- the ANYmal jump (stand, flight, stand; N = 40, PhaseBased grid, both events with
    switching-time optimisation) with the reference's SwitchingTimeOptimization over its minimum-dwell-time STOConstraints and the
    empty STOCostFunction of examples/anymal/python/jump_sto.py:104-108 in the loop (ocp_solver.cpp:119, 128-132, 143): the
    fixture also carries the event times before and after the iteration, the dwell-time rows and the switching-time directions.
    line_search="merit": the same with LineSearchMethod::MeritBacktracking (line_search.cpp:87-128): penalty parameter, directional
    derivative from a trial at step eps, Armijo backtracking; armijo = LineSearchSettings::armijo_control_rate (1.5 asks for more
    decrease than the linear model promises: every candidate is evaluated and rejected, the loop ends below min_step_size).
    line_search=True: SolverOptions::enable_line_search -- the reference's own LineSearch::computeStepSize (filter method,
    src/line_search/line_search.cpp:31-83) picks the primal step between computeStepSizes and integrateSolution
    (ocp_solver.cpp:133-139); every trial iterate's rigid-body quantities are injected like those of the iterate itself.
    robot="icub": BASELINE configs[3]'s robot -- iCub (nv = 35) on its two soles, SURFACE contacts (six rows per contact in f, mu and
    the switching constraint, Log6 placement residuals, frame rotations in the contact schedule), stand -> lift-off of both soles ->
    flight -> touch-down -> stand over eleven grid points, FrictionCone on the force part of the wrenches."""
    import ctypes as C
    from robotoc_amd import robot_model as rm
    from robotoc_amd.grid import (ANYMAL_Q_STANDING, ANYMAL_TROT_IMPACT_MASKS, ANYMAL_TROT_PHASE_MASKS, contact_masks)
    from robotoc_amd.types import GRID_IMPACT as GI, GRID_TERMINAL as GT
    icub = robot == "icub"
    assert not (icub and (sto or line_search))
    m = rm.load_named("icub" if icub else "anymal")
    nv, nq, nu, nc, cd = m.nv, m.nq, m.nv - 6, (2 if icub else 4), (6 if icub else 3)
    if icub:
        from robotoc_amd.grid import ICUB_Q_STANDING
        cs_ = jump_sto_sequence(ground_time=0.07, flying_time=0.06, nf=12)
        for e_ in cs_.events:
            e_.sto = False
        grids = discretize(10, 0.2, 0.0, cs_)
        masks = contact_masks(grids, [0b11, 0, 0b11], [0b11])
    elif sto:
        T_h, t_lift, t_land = 0.8, 0.31, 0.51
        grids = discretize(40, T_h, 0.0, jump_sto_sequence(ground_time=t_lift, flying_time=t_land - t_lift, nf=12), phase_based=True)
        masks = contact_masks(grids, [0b1111, 0, 0b1111], [0b1111])
    else:
        grids = discretize(10, 0.2, 0.0, anymal_trot_sequence(t0=0.03, swing=0.05, double_support=0.03, cycles=1))
        masks = contact_masks(grids, ANYMAL_TROT_PHASE_MASKS, ANYMAL_TROT_IMPACT_MASKS)
    n = len(grids)
    rng = np.random.default_rng(2468)
    qs = np.array(ICUB_Q_STANDING if icub else ANYMAL_Q_STANDING, dtype=float)
    feet = np.array([orc.rbd_contact_position(m, qs, c) for c in range(nc)])
    pos = np.tile(feet[None], (n, 1, 1)) + 0.003 * rng.uniform(-1, 1, (n, nc, 3))
    rot = None
    if icub:   # the soles' reference placements: the standing pose's, turned by a few hundredths of a radian
        R0 = [orc.rbd_contact_placement(m, qs, c)[0] for c in range(nc)]
        # (one placement per contact PHASE: FrictionCone reads ContactStatus::contactRotation, which a phase holds once -- the rigid-body
        # quantities are injected per grid point and would tolerate a placement of their own each, the cone would not)
        rot, ph = np.zeros((n, nc, 3, 3)), None
        for i, g in enumerate(grids):
            if ph is None or g.type in (GI, 2):
                ph = np.array([R0[c] @ orc.rbd_exp6(np.concatenate([np.zeros(3), 0.02 * rng.uniform(-1, 1, 3)]))[0] for c in range(nc)])
            rot[i] = ph
    rkw = lambda i: dict(rref=rot[i].reshape(nc, 9)) if icub else {}
    if sto:
        land = next(i for i, g in enumerate(grids) if g.type == GI)
        pos[land:, :, 0] += 0.1   # the feet land 10 cm ahead
    q = np.zeros((n, nq))
    for i in range(n):
        q[i] = qs
        q[i, :7] = orc.se3_integrate(qs[:7], 0.02 * rng.uniform(-1, 1, 6))
        q[i, 7:] += 0.05 * rng.uniform(-1, 1, nu)
    v, a, u = 0.2 * rng.uniform(-1, 1, (n, nv)), 0.5 * rng.uniform(-1, 1, (n, nv)), 5.0 * rng.uniform(-1, 1, (n, nu))
    if line_search:   # an iterate whose full Newton step is long (roomy slacks) and overshoots (fast joints): the filter has work to do
        v, a = 3.0 * rng.uniform(-1, 1, (n, nv)), 20.0 * rng.uniform(-1, 1, (n, nv))
    f = 5.0 * rng.uniform(-1, 1, (n, nc, cd))
    f[:, :, 2] = rng.uniform(60, 120, (n, nc))
    lmd, gmm, beta = (0.2 * rng.uniform(-1, 1, (n, nv)) for _ in range(3))
    mus, nup, xi = 0.2 * rng.uniform(-1, 1, (n, nc, cd)), 0.2 * rng.uniform(-1, 1, (n, 6)), 0.2 * rng.uniform(-1, 1, (n, cd * nc))
    mu = np.array([0.7, 0.6, 0.8, 0.5])[:nc]
    wq = np.concatenate([np.full(6, 10.0), np.full(nu, 1.0)])
    M = nv + 1
    cost = np.zeros((12, M))
    for k, val in ((0, qs), (3, wq), (4, np.full(nv, 1.0)), (5, np.full(nv, 1e-2)), (6, np.full(nu, 1e-2)), (7, 10.0 * wq), (8, np.full(nv, 1.0)),
                   (9, wq), (10, np.full(nv, 1.0)), (11, np.full(nv, 1e-2))):
        cost[k, :len(val)] = val
    limits = np.stack([np.full(nu, -1.8), np.full(nu, 1.8), np.full(nu, 4.0), np.full(nu, 60.0)])
    nrow = 6 * nu + 5 * nc
    slack, dual = rng.uniform(0.2, 2.0, (n, nrow)), rng.uniform(0.01, 0.3, (n, nrow))
    if line_search:
        limits = np.stack([np.full(nu, -6.0), np.full(nu, 6.0), np.full(nu, 40.0), np.full(nu, 400.0)])
        slack, dual = rng.uniform(20.0, 40.0, (n, nrow)), rng.uniform(1e-5, 1e-4, (n, nrow))
    barrier, tau = 1.0e-3, 0.995
    x0 = np.concatenate([orc.se3_integrate(qs[:7], 0.01 * rng.uniform(-1, 1, 6)), qs[7:] + 0.02 * rng.uniform(-1, 1, nu), 0.05 * rng.uniform(-1, 1, nv)])
    L = ref.lib()
    dp = C.POINTER(C.c_double)
    ptr = lambda x: np.ascontiguousarray(x, dtype=np.float64).ctypes.data_as(dp)
    plus = lambda qq, e: orc.rbd_integrate(m, qq, e)

    def sub(qf, q0):
        return np.concatenate([orc.se3_difference(q0[:7], qf[:7]), qf[7:] - q0[7:]])
    if icub:
        L.ref_ocp_begin_surface.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int]
        assert L.ref_ocp_begin_surface(nv, nu, nc, ptr(rot), n) == 0
    else:
        L.ref_ocp_begin(nv, nu, nc)

    def inject(key, arr):
        arr = np.asarray(arr, dtype=np.float64)
        keep = np.asfortranarray(arr.reshape(arr.shape[0], -1) if arr.size else arr.reshape(arr.shape[0], arr.shape[1] if arr.ndim > 1 else 1))
        assert L.ref_ocp_inject(key.encode(), keep.ctypes.data_as(dp), keep.shape[0], keep.shape[1]) == 0
    h = 2.0e-3
    for i, g in enumerate(grids):
        impact, terminal = g.type == GI, g.type == GT
        qi = q[i]
        qp = q[i - 1] if i > 0 else x0[:nq]
        # frame kinematics of every contact, popped by the stage's opening updateKinematics
        for c in range(nc):
            R = orc.rbd_contact_placement(m, qi, c)[0]
            dR = _richardson(lambda e: orc.rbd_contact_placement(m, plus(qi, e), c)[0].reshape(-1), nv)
            Jl = np.zeros((6, nv))
            for jj in range(nv):
                W = dR[:, jj].reshape(3, 3) @ R.T
                Jl[3:, jj] = R.T @ np.array([W[2, 1], W[0, 2], W[1, 0]])
            inject("frames", np.concatenate([R.reshape(-1), Jl.T.reshape(-1)]))
        inject("subtractConfiguration", sub(qi, qs))
        inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), qs), nv))
        if terminal:
            inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qp, plus(qi, e)), nv))
            continue
        qn = q[i + 1]
        inject("subtractConfiguration", sub(qi, qn))
        inject("dSubtractConfiguration_dqf", _richardson(lambda e: sub(plus(qi, e), qn), nv))
        inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qp, plus(qi, e)), nv))
        mask = int(masks[i])
        act = [c for c in range(nc) if (mask >> c) & 1]
        fstack = np.concatenate([f[i, c] for c in act]) if act else np.zeros(0)
        val = orc.rbd_eval(m, int(impact), qi, v[i], a[i], fstack, np.zeros(nu), mask, pos[i].reshape(-1), **rkw(i))
        J1 = orc.rbd_linearize_fd(m, int(impact), qi, v[i], a[i], fstack, np.zeros(nu), mask, pos[i].reshape(-1), eps=h, **rkw(i))
        J2 = orc.rbd_linearize_fd(m, int(impact), qi, v[i], a[i], fstack, np.zeros(nu), mask, pos[i].reshape(-1), eps=h / 2, **rkw(i))
        Jr = [(4.0 * np.asarray(J2[k]) - np.asarray(J1[k])) / 3.0 for k in range(3)]
        inject("ID", val[:nv])
        if impact:
            inject("impactVelocityResidual", val[nv:])
        else:
            inject("baumgarteResidual", val[nv:])
        inject("dIDdq", Jr[0][:nv]), inject("dIDdv", Jr[1][:nv]), inject("dIDda", Jr[2][:nv])
        if impact:
            inject("impactVelocity_dq", Jr[0][nv:]), inject("impactVelocity_dv", Jr[2][nv:])
        else:
            inject("baumgarte_dq", Jr[0][nv:]), inject("baumgarte_dv", Jr[1][nv:]), inject("baumgarte_da", Jr[2][nv:])
        if g.switching_constraint and not impact:
            dt1, dt2 = g.dt, grids[i + 1].dt
            dq_sw = (dt1 + dt2) * v[i] + dt1 * dt2 * a[i]
            q_plus = plus(qi, dq_sw)
            imp = [c for c in range(nc) if (int(masks[i + 2]) >> c) & 1]
            P_at = lambda qq: np.concatenate([orc.rbd_contact_position(m, qq, c) - pos[i + 2, c] for c in imp])
            if icub:   # SurfaceContact::computeContactPositionResidual (surface_contact.hxx:106-114): Log6(X_ref^-1 X)
                def P_at(qq):
                    out_ = []
                    for c in imp:
                        Rc, pc = orc.rbd_contact_placement(m, qq, c)
                        Rr = rot[i + 2, c]
                        out_.append(orc.rbd_log6(Rr.T @ Rc, Rr.T @ (pc - pos[i + 2, c])))
                    return np.concatenate(out_)
            inject("integrateConfiguration", q_plus)
            inject("contactPositionResidual", P_at(q_plus))
            inject("contactPositionDerivative", _richardson(lambda e: P_at(plus(q_plus, e)), nv))
            inject("dIntegrate_dq", _richardson(lambda e: sub(plus(plus(qi, e), dq_sw), q_plus), nv))
            inject("dIntegrate_dv", _richardson(lambda e: sub(plus(qi, dq_sw + e), q_plus), nv))
        inject("dSubtractConfiguration_dq0", _richardson(lambda e: sub(qi, plus(qn, e)), nv))
    inject("subtractConfiguration", sub(x0[:nq], q[0]))    # computeInitialStateDirection (state_equation.cpp:103)
    SL = L.ref_ocp_sol_len_cd(nv, nu, nc, cd)
    sol = np.zeros((n, SL))
    for i in range(n):
        mask = int(masks[i])
        fi, mi = np.zeros((nc, cd)), np.zeros((nc, cd))
        for c in range(nc):
            if (mask >> c) & 1:
                fi[c], mi[c] = f[i, c], mus[i, c]
        sol[i] = np.concatenate([q[i], v[i], a[i], u[i], fi.reshape(-1), lmd[i], gmm[i], beta[i], mi.reshape(-1), nup[i], xi[i]])
    from robotoc_amd.types import Grid
    garr = (Grid * n)(*grids)
    marr = np.ascontiguousarray(masks, dtype=np.uint32)
    out_dq, steps = np.zeros((n, nv)), np.zeros(6)
    sto_kw = {}
    if sto:
        min_dwell, sto_barrier, sto_tau, sto_reg = np.array([0.15, 0.15, 0.2]), 1.0e-3, 0.995, 1.0e-2
        ev_sto = (C.c_int * 2)(1, 1)
        L.ref_ocp_sto_setup.argtypes = [C.POINTER(C.c_int), C.c_int, dp, C.c_double, C.c_double, C.c_double, dp, dp]
        # the dwell-time rows away from their initialisation, so that residual and cmpl are not zero
        sto_slack, sto_dual = rng.uniform(0.05, 0.4, 3), rng.uniform(0.002, 0.05, 3)
        assert L.ref_ocp_sto_setup(ev_sto, 2, ptr(min_dwell), sto_barrier, sto_tau, sto_reg, ptr(sto_slack), ptr(sto_dual)) == 0
    L.ref_ocp_direction.argtypes = [C.POINTER(Grid), C.POINTER(C.c_uint), dp, C.c_int, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp, dp]
    rc = L.ref_ocp_direction(garr, marr.ctypes.data_as(C.POINTER(C.c_uint)), ptr(pos), n, ptr(mu), ptr(cost), ptr(limits), barrier, tau, ptr(x0[:nq]),
                             ptr(x0[nq:]), ptr(sol), ptr(slack), ptr(dual), out_dq.ctypes.data_as(dp), steps.ctypes.data_as(dp))
    assert rc == 0, rc

    def sto_result():
        et, con, ltq, perf, dts = np.zeros(2), np.zeros((6, 3)), np.zeros((2, 2)), np.zeros(2), np.zeros((n, 2))
        L.ref_ocp_sto_result.argtypes = [dp] * 5
        assert L.ref_ocp_sto_result(*[x.ctypes.data_as(dp) for x in (et, con, ltq, perf, dts)]) == 0
        return et, con, ltq, perf, dts
    if sto:
        et0, con_dir, ltq, perf, dts = sto_result()
        sto_kw = dict(sto_event_times=et0, sto_con_direction=con_dir, sto_lt_qtt=ltq, sto_perf=perf, sto_dts=dts, sto_min_dwell=min_dwell, sto_slack=sto_slack, sto_dual=sto_dual,
                      sto_scalars=np.array([sto_barrier, sto_tau, sto_reg, 0.0, T_h]))
    ls_kw = {}
    if line_search:
        # a filter that wants the violation halved: no trial below the (fraction-to-boundary) maximum step can deliver that, so every
        # candidate is evaluated and rejected and the loop ends below min_step_size -- the whole backtracking path runs
        rate, min_step, cost_rate, viol_rate = 0.75, 0.05, 0.005, 0.5
        merit = line_search == "merit"
        margin, eps = 0.05, 1.0e-8   # LineSearchSettings' defaults; armijo_control_rate: the default 0.001, or the caller's (see FIXTURES)
        L.ref_ocp_trial_solution.argtypes = [C.c_double, dp, dp]
        L.ref_ocp_line_search.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, dp]
        L.ref_ocp_line_search_merit.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, dp]
        trials, alpha = [], float(steps[0])
        if merit:   # the directional derivative's trial at step eps comes first (line_search.cpp:98-103)
            q_tr = np.array([plus(q[i], eps * out_dq[i]) for i in range(n)])
            s_tr = np.zeros((n, SL))
            assert L.ref_ocp_trial_solution(eps, ptr(q_tr), s_tr.ctypes.data_as(dp)) == 0
            trials.append((eps, q_tr, s_tr))
        while alpha > min_step:
            q_tr = np.array([plus(q[i], alpha * out_dq[i]) for i in range(n)])
            s_tr = np.zeros((n, SL))
            assert L.ref_ocp_trial_solution(alpha, ptr(q_tr), s_tr.ctypes.data_as(dp)) == 0
            trials.append((alpha, q_tr, s_tr))
            alpha *= rate
        o_v, o_a, o_f = nq, nq + nv, nq + 2 * nv + nu
        L.ref_ocp_trial_eval.argtypes = [C.c_double, dp]
        trial_evals = []
        for k_pass, (alpha, q_tr, s_tr) in enumerate(trials + trials):   # once each on its own (the values), then all for the line search
            if k_pass == len(trials):
                pass
            for i in range(n):   # dms_trial_.integratePrimalSolution
                inject("integrateConfiguration", q_tr[i])
            for i, g in enumerate(grids):   # dms_trial_.evalOCP, stage by stage
                impact, terminal = g.type == GI, g.type == GT
                qi, vi, ai = s_tr[i, :nq], s_tr[i, o_v:o_v + nv], s_tr[i, o_a:o_a + nv]
                for c in range(nc):   # updateKinematics: the cones read the frame rotation; no Jacobian in an evaluation
                    inject("frames", np.concatenate([orc.rbd_contact_placement(m, qi, c)[0].reshape(-1), np.zeros(6 * nv)]))
                inject("subtractConfiguration", sub(qi, qs))
                if terminal:
                    continue
                inject("subtractConfiguration", sub(qi, s_tr[i + 1, :nq]))
                mask = int(masks[i])
                act = [c for c in range(nc) if (mask >> c) & 1]
                fstack = np.concatenate([s_tr[i, o_f + 3 * c:o_f + 3 * c + 3] for c in act]) if act else np.zeros(0)
                val = orc.rbd_eval(m, int(impact), qi, vi, ai, fstack, np.zeros(nu), mask, pos[i].reshape(-1))
                inject("ID", val[:nv])
                inject("impactVelocityResidual" if impact else "baumgarteResidual", val[nv:])
                if g.switching_constraint and not impact:
                    dt1, dt2 = g.dt, grids[i + 1].dt
                    q_plus = plus(qi, (dt1 + dt2) * vi + dt1 * dt2 * ai)
                    imp = [c for c in range(nc) if (int(masks[i + 2]) >> c) & 1]
                    inject("integrateConfiguration", q_plus)
                    inject("contactPositionResidual", np.concatenate([orc.rbd_contact_position(m, q_plus, c) - pos[i + 2, c] for c in imp]))
            if k_pass < len(trials):
                ev = np.zeros(3)
                assert L.ref_ocp_trial_eval(alpha, ev.ctypes.data_as(dp)) == 0
                trial_evals.append(ev)
        ls_step = np.zeros(2)
        if merit:
            assert L.ref_ocp_line_search_merit(rate, min_step, armijo, margin, eps, ls_step.ctypes.data_as(dp)) == 0
        else:
            assert L.ref_ocp_line_search(rate, min_step, cost_rate, viol_rate, ls_step.ctypes.data_as(dp)) == 0
        n_eval = len(trials) - int(round(ls_step[1])) // (n - 1)
        print("  line search: max primal step %.4f -> accepted %.4f (%d of %d candidate trials evaluated), eval0 cost %.4e barrier %.4e violation %.4e"
              % (steps[0], ls_step[0], n_eval, len(trials), steps[3], steps[4], steps[5]))
        ls_kw = dict(ls_step=ls_step[:1], ls_trials_evaluated=np.array([n_eval]), ls_max_step=steps[:1].copy(), ls_trial_evals=np.array(trial_evals), ls_eval0=steps[3:6].copy(), ls_settings=np.array([rate, min_step, cost_rate, viol_rate]),
                     ls_trial_steps=np.array([t[0] for t in trials]))
        if merit:
            # LineSearch::penaltyParam restated on the iterate (line_search.cpp:120-128; SplitSolution::lagrangeMultiplierLinfNorm,
            # split_solution.cpp:126-134) and the merit values the reference's decisions rest on, for the test's report
            o_l = nq + 2 * nv + nu + 3 * nc
            pen = 0.0
            for i, g in enumerate(grids):
                act = [c for c in range(nc) if (int(masks[i]) >> c) & 1]
                vals = [np.abs(sol[i, o_l:o_l + 2 * nv]).max()]                      # lmd, gmm
                if g.type != GT:
                    vals.append(np.abs(sol[i, o_l + 2 * nv:o_l + 3 * nv]).max())     # beta
                    vals.append(np.abs(sol[i, o_l + 3 * nv + 3 * nc:o_l + 3 * nv + 3 * nc + 6]).max())   # nu_passive
                    if act:
                        vals.append(max(np.abs(sol[i, o_l + 3 * nv + 3 * c:o_l + 3 * nv + 3 * c + 3]).max() for c in act))   # mu of the active contacts
                    if g.switching_constraint and g.type != GI:
                        vals.append(np.abs(sol[i, o_l + 3 * nv + 3 * nc + 6:o_l + 3 * nv + 3 * nc + 6 + g.dims]).max())      # xi
                pen = max(pen, max(vals))
            ls_kw.update(ls_merit_settings=np.array([armijo, margin, eps]), ls_penalty=np.array([pen * (1.0 + margin)]))
        steps[0] = ls_step[0]
    q_int = np.array([plus(q[i], steps[0] * out_dq[i]) for i in range(n)])
    sol_out, slack_out, dual_out = np.zeros((n, SL)), np.zeros((n, nrow)), np.zeros((n, nrow))
    L.ref_ocp_integrate.argtypes = [dp, dp, dp, dp]
    rc = L.ref_ocp_integrate(ptr(q_int), sol_out.ctypes.data_as(dp), slack_out.ctypes.data_as(dp), dual_out.ctypes.data_as(dp))
    assert rc == 0, rc
    if sto:
        et1, con_out = sto_result()[:2]
        sto_kw.update(sto_event_times_out=et1, sto_con_out=con_out)
        print("  event times %s -> %s, dts %s, STO kkt %.3e" % (et0, et1, dts[[i for i, g in enumerate(grids) if g.type in (1, 2)], 0], perf[0]))
    np.savez_compressed(os.path.join(HERE, name), sol_in=sol, sol_out=sol_out, slack=slack, dual=dual, slack_out=slack_out, dual_out=dual_out,
                        steps=steps, x0=x0, pos=pos, mu=mu, cost=cost, limits=limits, masks=masks, scalars=np.array([barrier, tau]),
                        **(dict(rot=rot, robot=np.array("icub")) if icub else {}), **grid_table(grids), **sto_kw, **ls_kw)
    print(name, "n %d, KKT error %.6e, steps %.4f / %.4f" % (n, steps[2], steps[0], steps[1]))


FIXTURES = {
    "riccati": main,
    "unconstr_solver": lambda: (unconstr_solver_fixture("ref_iiwa14_unconstr_solver.npz", False),
                                unconstr_solver_fixture("ref_iiwa14_unconstr_solver_limits.npz", True)),
    "unconstr_line_search": lambda: [unconstr_line_search_fixture("ref_iiwa14_unconstr_line_search%s.npz" % tag, lim, back)
                                     for tag, lim, back in (("", False, False), ("_limits", True, False), ("_backtrack", False, True),
                                                            ("_limits_backtrack", True, True))],
    "contact_stage": contact_stage_fixture,
    "impact_terminal_stage": impact_and_terminal_stage_fixture,
    "icub_surface_stage": icub_surface_stage_fixture,
    "ocp_iteration": ocp_solver_iteration_fixture,
    "riccati_full_size": full_size_fixtures,
    "riccati_icub32": icub32_full_size,
    "ocp_iteration_icub": lambda: ocp_solver_iteration_fixture("ref_icub_ocp_solver_iteration.npz", robot="icub"),
    "ocp_iteration_sto": lambda: ocp_solver_iteration_fixture("ref_anymal_jump_sto_solver_iteration.npz", sto=True),
    "ocp_iteration_line_search": lambda: ocp_solver_iteration_fixture("ref_anymal_ocp_solver_iteration_line_search.npz", line_search=True),
    "ocp_iteration_line_search_merit": lambda: (ocp_solver_iteration_fixture("ref_anymal_ocp_solver_iteration_line_search_merit.npz", line_search="merit"),
                                                ocp_solver_iteration_fixture("ref_anymal_ocp_solver_iteration_line_search_merit_reject.npz", line_search="merit", armijo=1.5)),
}

if __name__ == "__main__":   # python tests/golden/make_ref_golden.py [fixture ...]   (default: all)
    ref.build()
    for key in (sys.argv[1:] or list(FIXTURES)):
        FIXTURES[key]()
