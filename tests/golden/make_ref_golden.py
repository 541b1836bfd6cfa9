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


if __name__ == "__main__":
    main()
    unconstr_solver_fixture("ref_iiwa14_unconstr_solver.npz", False)
    unconstr_solver_fixture("ref_iiwa14_unconstr_solver_limits.npz", True)
