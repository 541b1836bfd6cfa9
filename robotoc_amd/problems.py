"""Deterministic synthetic workloads for the hot path.

The reference cannot produce stage data here (Pinocchio is absent), so inputs are
synthetic but structurally faithful re-creations of the reference's own test
factories:
  * condensed KKT stages  -- test/test_helper/kkt_factory.cpp:7-64
  * Riccati factorisation -- test/test_helper/riccati_factory.cpp:9-17
  * pre-condensation data -- SplitKKTMatrix::setRandom (src/core/split_kkt_matrix.cpp:153-178)
                             and test/dynamics/contact_dynamics_test.cpp:87-120
Everything is seeded: seed = BASE_SEED + instance index.
"""
import numpy as np

from .grid import (ContactSequence, anymal_trot_sequence, discretize, jump_sto_sequence,
                   uniform_grid)
from .types import (GRID_IMPACT, GRID_TERMINAL, Records, anymal_dims, icub_dims, iiwa14_dims)

BASE_SEED = 20260925


def _rnd(rng, *shape):
    """Eigen::MatrixXd::Random analogue: uniform in [-1, 1]."""
    return rng.uniform(-1.0, 1.0, size=shape)


def _spd(rng, n, shift=0.0):
    s = _rnd(rng, n, n)
    return s @ s.T + shift * np.eye(n)


def fill_kkt_instance(L, grids, kkt, rng, mode="dynamics", floating_base=None):
    """Fill one instance's condensed KKT records ([stages, stride]).

    mode="factory":  Fvq, Fvv, Fvu ~ U[-1,1] exactly like kkt_factory.cpp:21-23.
    mode="dynamics": Fvq = dt*R, Fvv = I + dt*R, Fvu = dt*R (the magnitudes
                     condenseContactDynamics produces, contact_dynamics.cpp:130-134),
                     which keeps a 40-stage recursion well conditioned.
    """
    d = L.dims
    nv, nu, nx, ns = d.nv, d.nu, 2 * d.nv, d.ns_max
    K = Records(L, "kkt")
    fb = (d.np > 0) if floating_base is None else floating_base
    for i, g in enumerate(grids):
        rec = kkt[i]
        if g.type == GRID_TERMINAL:
            K.f(rec, "Qxx")[...] = _spd(rng, nx)
            K.f(rec, "lx")[...] = _rnd(rng, nx)
            continue
        dt = g.dt if g.dt > 0 else 0.0
        Fxx = K.f(rec, "Fxx")
        Fxx[...] = 0.0
        Fxx[:nv, :nv] = np.eye(nv)
        if g.type != GRID_IMPACT:
            Fxx[:nv, nv:] = dt * np.eye(nv)
        if fb:
            Fxx[:6, :6] = _rnd(rng, 6, 6)
            if g.type != GRID_IMPACT:
                Fxx[:6, nv:nv + 6] = _rnd(rng, 6, 6) * (dt if mode == "dynamics" else 1.0)
        if mode == "factory":
            Fxx[nv:, :nv] = _rnd(rng, nv, nv)
            Fxx[nv:, nv:] = _rnd(rng, nv, nv)
        else:
            sc = dt if g.type != GRID_IMPACT else 0.1
            Fxx[nv:, :nv] = sc * _rnd(rng, nv, nv)
            Fxx[nv:, nv:] = np.eye(nv) + sc * _rnd(rng, nv, nv)
        K.f(rec, "Fx")[...] = _rnd(rng, nx)
        K.f(rec, "lx")[...] = _rnd(rng, nx)
        if g.type == GRID_IMPACT:
            K.f(rec, "Qxx")[...] = _spd(rng, nx)
            continue
        K.f(rec, "Fvu")[...] = _rnd(rng, nv, nu) * (dt if mode == "dynamics" else 1.0)
        H = _spd(rng, nx + nu)
        K.f(rec, "Qxx")[...] = H[:nx, :nx]
        K.f(rec, "Qxu")[...] = H[:nx, nx:]
        K.f(rec, "Quu")[...] = H[nx:, nx:]
        K.f(rec, "lu")[...] = _rnd(rng, nu)
        # STO terms (used only on sto grids but always filled, like CreateSplitKKTMatrix)
        K.f(rec, "fx")[...] = _rnd(rng, nx)
        K.f(rec, "hx")[...] = _rnd(rng, nx)
        K.f(rec, "hu")[...] = _rnd(rng, nu)
        scal = K.f(rec, "scal")
        qtt = abs(rng.uniform(-1, 1)) + 0.1
        scal[0] = qtt           # Qtt > 0
        scal[1] = -qtt          # Qtt_prev = -Qtt (intermediate_stage.cpp:145)
        scal[2] = rng.uniform(-1, 1)  # h
        if g.dims > 0:
            m = g.dims
            K.f(rec, "Phix")[:m, :] = _rnd(rng, m, nx)
            K.f(rec, "Phiu")[:m, :] = _rnd(rng, m, nu)
            K.f(rec, "Phit")[:m] = _rnd(rng, m)
            K.f(rec, "Pres")[:m] = _rnd(rng, m)


def make_kkt_batch(L, grids, batch, mode="dynamics", first_instance=0):
    K = Records(L, "kkt")
    kkt = K.zeros(batch, len(grids))
    for b in range(batch):
        rng = np.random.default_rng(BASE_SEED + first_instance + b)
        fill_kkt_instance(L, grids, kkt[b], rng, mode=mode)
    return kkt


def make_kkt_batch_tiled(L, grids, batch, unique=8, mode="dynamics", first_instance=0):
    """Bench helper: `unique` distinct instances tiled to `batch` (filling 4096
    instances record by record in numpy takes minutes; the arithmetic does not care)."""
    u = min(unique, batch)
    base = make_kkt_batch(L, grids, u, mode=mode, first_instance=first_instance)
    reps = (batch + u - 1) // u
    return np.ascontiguousarray(np.tile(base, (reps, 1, 1))[:batch])


class _NumpyGen:
    """Random source + the few array helpers the vectorised generators need, numpy flavour."""

    def __init__(self, key, device=None):
        self.rng = np.random.default_rng([BASE_SEED] + list(key))

    def U(self, *shape):
        return self.rng.uniform(-1.0, 1.0, size=shape)

    def zeros(self, *shape):
        return np.zeros(shape)

    eye = staticmethod(np.eye)
    tril = staticmethod(np.tril)
    abs = staticmethod(np.abs)

    @staticmethod
    def diag(A):  # writable view of the diagonals of a stack of square matrices
        return np.einsum("...ii->...i", A)


class _TorchGen:
    """The same on a torch device (bench.py: 4096 instances are generated in HBM in well under a second)."""

    def __init__(self, key, device="cuda"):
        import torch
        self.t = torch
        self.dev = device
        self.g = torch.Generator(device=device)
        self.g.manual_seed(int(np.random.SeedSequence([BASE_SEED] + list(key)).generate_state(1, np.uint64)[0] >> 1))

    def U(self, *shape):
        return self.t.rand(shape, generator=self.g, dtype=self.t.float64, device=self.dev) * 2.0 - 1.0

    def zeros(self, *shape):
        return self.t.zeros(shape, dtype=self.t.float64, device=self.dev)

    def eye(self, n):
        return self.t.eye(n, dtype=self.t.float64, device=self.dev)

    def tril(self, a):
        return self.t.tril(a)

    def abs(self, a):
        return self.t.abs(a)

    @staticmethod
    def diag(A):
        return A.diagonal(dim1=-2, dim2=-1)


def _gen(backend, key, device):
    return (_TorchGen if backend == "torch" else _NumpyGen)(key, device)


def _spd_b(G, b, n):
    s = G.U(b, n, n)
    return s @ s.swapaxes(-1, -2)


def make_kkt_batch_unique(L, grids, batch, mode="dynamics", seed=0, chunk=256, out=None, backend="numpy",
                          device="cuda"):
    """`batch` DISTINCT instances with the statistics of fill_kkt_instance, generated stage by stage for a
    whole chunk of instances at once (vectorised over the instance axis; backend="torch" generates straight
    into HBM).  Chunk c draws from the stream keyed (BASE_SEED, seed, c); the values differ from
    make_kkt_batch's per-instance streams, the distribution does not (BASELINE config 5: "randomised"
    instances).  out: a zeroed [batch, stages, stride] array / tensor to fill."""
    d = L.dims
    nv, nu, nx = d.nv, d.nu, 2 * d.nv
    K = Records(L, "kkt")
    if backend == "torch":
        chunk = batch
    kkt = out if out is not None else _gen(backend, (seed,), device).zeros(batch, len(grids), K.stride)
    fb = d.np > 0
    for c0 in range(0, batch, chunk):
        nb = min(chunk, batch - c0)
        G = _gen(backend, (seed, c0 // chunk), device)
        U = lambda *shape: G.U(nb, *shape)
        eye_v = G.eye(nv)
        blk = kkt[c0:c0 + nb]
        for i, g in enumerate(grids):
            rec = blk[:, i]
            if g.type == GRID_TERMINAL:
                K.f(rec, "Qxx")[...] = _spd_b(G, nb, nx)
                K.f(rec, "lx")[...] = U(nx)
                continue
            dt = g.dt if g.dt > 0 else 0.0
            imp = g.type == GRID_IMPACT
            Fxx = K.f(rec, "Fxx")
            Fxx[...] = 0.0
            Fxx[:, :nv, :nv] = eye_v
            if not imp:
                Fxx[:, :nv, nv:] = dt * eye_v
            if fb:
                Fxx[:, :6, :6] = U(6, 6)
                if not imp:
                    Fxx[:, :6, nv:nv + 6] = U(6, 6) * (dt if mode == "dynamics" else 1.0)
            if mode == "factory":
                Fxx[:, nv:, :nv] = U(nv, nv)
                Fxx[:, nv:, nv:] = U(nv, nv)
            else:
                sc = dt if not imp else 0.1
                Fxx[:, nv:, :nv] = sc * U(nv, nv)
                Fxx[:, nv:, nv:] = eye_v + sc * U(nv, nv)
            K.f(rec, "Fx")[...] = U(nx)
            K.f(rec, "lx")[...] = U(nx)
            if imp:
                K.f(rec, "Qxx")[...] = _spd_b(G, nb, nx)
                continue
            K.f(rec, "Fvu")[...] = U(nv, nu) * (dt if mode == "dynamics" else 1.0)
            H = _spd_b(G, nb, nx + nu)
            K.f(rec, "Qxx")[...] = H[:, :nx, :nx]
            K.f(rec, "Qxu")[...] = H[:, :nx, nx:]
            K.f(rec, "Quu")[...] = H[:, nx:, nx:]
            K.f(rec, "lu")[...] = U(nu)
            K.f(rec, "fx")[...] = U(nx)
            K.f(rec, "hx")[...] = U(nx)
            K.f(rec, "hu")[...] = U(nu)
            scal = K.f(rec, "scal")
            qtt = G.abs(U()) + 0.1
            scal[:, 0] = qtt
            scal[:, 1] = -qtt
            scal[:, 2] = U()
            if g.dims > 0:
                m = g.dims
                K.f(rec, "Phix")[:, :m, :] = U(m, nx)
                K.f(rec, "Phiu")[:, :m, :] = U(m, nu)
                K.f(rec, "Phit")[:, :m] = U(m)
                K.f(rec, "Pres")[:, :m] = U(m)
    return kkt


def make_dx0_unique(L, batch, seed=0, scale=0.1, backend="numpy", device="cuda"):
    """`batch` distinct initial state directions ~ U[-scale, scale] from one stream."""
    return scale * _gen(backend, (7919, seed), device).U(batch, 2 * L.dims.nv)


def make_dx0(L, batch, first_instance=0, scale=0.1):
    """d[0].dx = (q0 - q, v0 - v) ~ U[-0.1, 0.1]."""
    out = np.zeros((batch, 2 * L.dims.nv))
    for b in range(batch):
        rng = np.random.default_rng(BASE_SEED + 7919 + first_instance + b)
        out[b] = scale * _rnd(rng, 2 * L.dims.nv)
    return out


def fill_unconstr_instance(L, nstages, kkt, rng):
    """test/riccati/unconstr_riccati_recursion_test.cpp:37-45: SPD [Qxx Qxu; . Qaa], random Fx, lx, la.
    Qaa lives in the Quu slot, la in the lu slot."""
    d = L.dims
    nv, nx = d.nv, 2 * d.nv
    K = Records(L, "kkt")
    for i in range(nstages):
        rec = kkt[i]
        if i == nstages - 1:
            K.f(rec, "Qxx")[...] = _spd(rng, nx)
            K.f(rec, "lx")[...] = _rnd(rng, nx)
            continue
        H = _spd(rng, nx + nv)
        K.f(rec, "Qxx")[...] = H[:nx, :nx]
        K.f(rec, "Qxu")[...] = H[:nx, nx:]
        K.f(rec, "Quu")[...] = H[nx:, nx:]
        K.f(rec, "Fx")[...] = _rnd(rng, nx)
        K.f(rec, "lx")[...] = _rnd(rng, nx)
        K.f(rec, "lu")[...] = _rnd(rng, nv)


# ---- named configurations of BASELINE.json ---------------------------------------
def config_iiwa14():
    """configs[0]: iiwa14 UnconstrOCPSolver, nv=7, N=20, T=1 (examples/iiwa14/unconstr_ocp_benchmark.cpp:59-60)."""
    dims = iiwa14_dims()
    return dims, uniform_grid(20, 1.0 / 20), dict(name="iiwa14_unconstr", dt=0.05)


def config_anymal_trot(N=40, dt=0.02):
    """configs[1]: ANYmal trot, nv=18, 4 point contacts, N=40; 2 lifts + 2 impacts -> 47 grids."""
    dims = anymal_dims()
    cs = anymal_trot_sequence(t0=0.11, swing=0.2, double_support=0.1, cycles=1)
    return dims, discretize(N, N * dt, 0.0, cs), dict(name="anymal_trot")


def config_anymal_jump_sto(N=40, dt=0.02):
    """configs[2]: ANYmal jump with switching-time optimisation, 3 phases, ns=12."""
    dims = anymal_dims()
    cs = jump_sto_sequence(ground_time=0.31, flying_time=0.2, nf=12)
    return dims, discretize(N, N * dt, 0.0, cs, phase_based=True), dict(name="anymal_jump_sto")


def config_icub_jump(N=30, dt=0.02, nv=35):
    """configs[3]: iCub jump, 2 surface contacts (nf=12), stand-flight-stand."""
    dims = icub_dims(nv)
    cs = jump_sto_sequence(ground_time=0.21, flying_time=0.2, nf=12)
    for e in cs.events:
        e.sto = False
    return dims, discretize(N, N * dt, 0.0, cs), dict(name="icub%d_jump" % nv)


# ---- pre-condensation stage data (inputs of condenseContactDynamics / condenseImpactDynamics) ----
def fill_precondense_instance(L, grids, kkt, cdd, rng):
    """One instance: the un-condensed KKT pieces a robotoc stage holds right before
    condenseContactDynamics (intermediate_stage.cpp:134-136), synthetic but structurally faithful
    (SURVEY 8d): M = dIDda SPD (L L^T + I), J = dCda random full row rank, dIDCdqv / IDC random,
    Qaa diagonal positive, Qff SPD, Qqf random, cost Hessians SPD, state equation blocks as
    linearizeStateEquation leaves them (Fqq = I with a 6x6 corner, Fqv = dt I)."""
    d = L.dims
    nv, nu, nx, npv = d.nv, d.nu, 2 * d.nv, d.np
    K, Cd = Records(L, "kkt"), Records(L, "cdd")
    for i, g in enumerate(grids):
        kr, cr = kkt[i], cdd[i]
        if g.type == GRID_TERMINAL:
            K.f(kr, "Qxx")[...] = _spd(rng, nx)
            K.f(kr, "lx")[...] = _rnd(rng, nx)
            continue
        nf, ns = g.dimf, g.dims
        nvf = nv + nf
        impact = g.type == GRID_IMPACT
        dt = g.dt
        A = K.f(kr, "Fxx")
        A[...] = 0.0
        A[:nv, :nv] = np.eye(nv)
        if npv > 0:
            A[:6, :6] = _rnd(rng, 6, 6)
        if not impact:
            A[:nv, nv:] = dt * np.eye(nv)
        K.f(kr, "Fx")[...] = _rnd(rng, nx)
        K.f(kr, "lx")[...] = _rnd(rng, nx)
        K.f(kr, "Qxx")[...] = _spd(rng, nx)
        if not impact:
            K.f(kr, "Qxu")[...] = 0.1 * _rnd(rng, nx, nu)
            K.f(kr, "Quu")[...] = np.diag(np.abs(_rnd(rng, nu)) + 0.1)
            K.f(kr, "lu")[...] = _rnd(rng, nu)
            K.f(kr, "hx")[...] = _rnd(rng, nx)
            K.f(kr, "hu")[...] = _rnd(rng, nu)
            K.f(kr, "fx")[...] = _rnd(rng, nx)
            sc = K.f(kr, "scal")
            sc[0] = abs(rng.uniform(-1, 1)) + 0.1
            sc[1] = -sc[0]
            sc[2] = rng.uniform(-1, 1)
            if ns > 0:
                K.f(kr, "Phix")[:ns] = _rnd(rng, ns, nx)
                K.f(kr, "Phit")[:ns] = _rnd(rng, ns)
                K.f(kr, "Pres")[:ns] = _rnd(rng, ns)
                Cd.f(cr, "Phia")[:ns] = _rnd(rng, ns, nv)
        Lm = np.tril(_rnd(rng, nv, nv))
        Cd.f(cr, "dIDda")[...] = Lm @ Lm.T + np.eye(nv)
        D = Cd.f(cr, "dIDCdqv")
        D[:nvf, :] = _rnd(rng, nvf, nx)
        if impact:
            D[:nv, nv:] = 0.0  # RNEAImpactDerivatives has no velocity block (impact_dynamics.cpp:44-50)
        elif nf > 0:
            Cd.f(cr, "dCda")[:nf] = _rnd(rng, nf, nv)
        Cd.f(cr, "IDC")[:nvf] = _rnd(rng, nvf)
        Cd.f(cr, "Qaa")[...] = np.abs(_rnd(rng, nv)) + 0.1
        if nf > 0:
            Cd.f(cr, "Qff")[:nf, :nf] = _spd(rng, nf)
            Cd.f(cr, "Qqf")[:, :nf] = _rnd(rng, nv, nf)
            Cd.f(cr, "lf")[:nf] = _rnd(rng, nf)
            Cd.f(cr, "hf")[:nf] = _rnd(rng, nf)
        Cd.f(cr, "la")[...] = _rnd(rng, nv)
        Cd.f(cr, "ha")[...] = _rnd(rng, nv)
        if npv > 0:
            Cd.f(cr, "lu_passive")[:npv] = _rnd(rng, npv)


def make_precondense_batch(L, grids, batch, first_instance=0):
    kkt = Records(L, "kkt").zeros(batch, len(grids))
    cdd = Records(L, "cdd").zeros(batch, len(grids))
    for b in range(batch):
        rng = np.random.default_rng(BASE_SEED + 104729 + first_instance + b)
        fill_precondense_instance(L, grids, kkt[b], cdd[b], rng)
    return kkt, cdd


def make_precondense_batch_unique(L, grids, batch, seed=0, chunk=128, backend="numpy", device="cuda", out=None):
    """`batch` DISTINCT instances with the statistics of fill_precondense_instance, vectorised over the
    instance axis (see make_kkt_batch_unique).  out: zeroed (kkt, cdd) arrays / tensors to fill."""
    d = L.dims
    nv, nu, nx, npv = d.nv, d.nu, 2 * d.nv, d.np
    K, Cd = Records(L, "kkt"), Records(L, "cdd")
    if backend == "torch":
        chunk = batch
    G0 = _gen(backend, (104729, seed), device)
    kkt, cdd = out if out is not None else (G0.zeros(batch, len(grids), K.stride), G0.zeros(batch, len(grids), Cd.stride))
    for c0 in range(0, batch, chunk):
        nb = min(chunk, batch - c0)
        G = _gen(backend, (104729, seed, c0 // chunk), device)
        U = lambda *shape: G.U(nb, *shape)
        eye_v = G.eye(nv)
        for i, g in enumerate(grids):
            kr, cr = kkt[c0:c0 + nb, i], cdd[c0:c0 + nb, i]
            if g.type == GRID_TERMINAL:
                K.f(kr, "Qxx")[...] = _spd_b(G, nb, nx)
                K.f(kr, "lx")[...] = U(nx)
                continue
            nf, ns = g.dimf, g.dims
            nvf = nv + nf
            impact = g.type == GRID_IMPACT
            A = K.f(kr, "Fxx")
            A[:, :nv, :nv] = eye_v
            if npv > 0:
                A[:, :6, :6] = U(6, 6)
            if not impact:
                A[:, :nv, nv:] = g.dt * eye_v
            K.f(kr, "Fx")[...] = U(nx)
            K.f(kr, "lx")[...] = U(nx)
            K.f(kr, "Qxx")[...] = _spd_b(G, nb, nx)
            if not impact:
                K.f(kr, "Qxu")[...] = 0.1 * U(nx, nu)
                G.diag(K.f(kr, "Quu"))[...] = G.abs(U(nu)) + 0.1
                K.f(kr, "lu")[...] = U(nu)
                K.f(kr, "hx")[...] = U(nx)
                K.f(kr, "hu")[...] = U(nu)
                K.f(kr, "fx")[...] = U(nx)
                sc = K.f(kr, "scal")
                sc[:, 0] = G.abs(U()) + 0.1
                sc[:, 1] = -sc[:, 0]
                sc[:, 2] = U()
                if ns > 0:
                    K.f(kr, "Phix")[:, :ns] = U(ns, nx)
                    K.f(kr, "Phit")[:, :ns] = U(ns)
                    K.f(kr, "Pres")[:, :ns] = U(ns)
                    Cd.f(cr, "Phia")[:, :ns] = U(ns, nv)
            Lm = G.tril(U(nv, nv))
            Cd.f(cr, "dIDda")[...] = Lm @ Lm.swapaxes(-1, -2) + eye_v
            D = Cd.f(cr, "dIDCdqv")
            D[:, :nvf, :] = U(nvf, nx)
            if impact:
                D[:, :nv, nv:] = 0.0
            elif nf > 0:
                Cd.f(cr, "dCda")[:, :nf] = U(nf, nv)
            Cd.f(cr, "IDC")[:, :nvf] = U(nvf)
            Cd.f(cr, "Qaa")[...] = G.abs(U(nv)) + 0.1
            if nf > 0:
                Cd.f(cr, "Qff")[:, :nf, :nf] = _spd_b(G, nb, nf)
                Cd.f(cr, "Qqf")[:, :, :nf] = U(nv, nf)
                Cd.f(cr, "lf")[:, :nf] = U(nf)
                Cd.f(cr, "hf")[:, :nf] = U(nf)
            Cd.f(cr, "la")[...] = U(nv)
            Cd.f(cr, "ha")[...] = U(nv)
            if npv > 0:
                Cd.f(cr, "lu_passive")[:, :npv] = U(npv)
    return kkt, cdd


def make_constraint_batch_unique(L, grids, batch, barrier=1.0e-3, seed=0, backend="numpy", device="cuda", out=None):
    """make_constraint_batch for `batch` distinct instances from one stream."""
    N = Records(L, "con")
    G = _gen(backend, (15485863, seed), device)
    con = out if out is not None else G.zeros(batch, len(grids), N.stride)
    shp = (batch, len(grids), L.dims.nc_max)
    slack = G.abs(G.U(*shp)) + 0.05
    dual = barrier / slack * (1.0 + 0.3 * G.U(*shp))
    N.f(con, "slack")[...] = slack
    N.f(con, "dual")[...] = dual
    N.f(con, "residual")[...] = 0.01 * G.U(*shp)
    N.f(con, "cmpl")[...] = slack * dual - barrier
    return con


def make_cone_batch_unique(L, grids, batch, max_contacts, seed=0, backend="numpy", device="cuda"):
    """make_cone_batch for `batch` distinct instances from one stream."""
    from .types import cone_dgdf_off, cone_stride
    nv = L.dims.nv
    cs, off = cone_stride(nv, max_contacts), cone_dgdf_off(nv, max_contacts)
    n = len(grids)
    G = _gen(backend, (7919, 1, seed), device)
    out = G.zeros(batch, n, cs)
    mu = 0.7 / np.sqrt(2.0)
    cone = G.zeros(5, 3)
    for r, row in enumerate([[0, 0, -1.0], [1, 0, -mu], [-1, 0, -mu], [0, 1, -mu], [0, -1, -mu]]):
        for c_, v in enumerate(row):
            cone[r, c_] = v
    for k in range(max_contacts):
        dq = 0.3 * G.U(batch, n, 5, nv)
        a = 0.3 * G.U(batch, n, 3)
        Rm = G.zeros(batch, n, 3, 3)
        Rm[..., 0, 1], Rm[..., 0, 2] = -a[..., 2], a[..., 1]
        Rm[..., 1, 0], Rm[..., 1, 2] = a[..., 2], -a[..., 0]
        Rm[..., 2, 0], Rm[..., 2, 1] = -a[..., 1], a[..., 0]
        Rm += G.eye(3)
        df = cone @ Rm.swapaxes(-1, -2)                     # [batch, n, 5, 3]
        out[..., k * 5 * nv:(k + 1) * 5 * nv] = dq.swapaxes(-1, -2).reshape(batch, n, -1)
        out[..., off + k * 15:off + (k + 1) * 15] = df.swapaxes(-1, -2).reshape(batch, n, -1)
    return out


def make_constraint_batch(L, grids, batch, barrier=1.0e-3, first_instance=0):
    """ConstraintComponentData of the joint-limit rows right after linearizeConstraints:
    slack > 0, dual = barrier / slack (perturbed, as after a few interior-point iterations),
    residual = g + slack, cmpl = slack * dual - barrier (pdipm.hxx:13-40, barrier as in
    examples/anymal/trot.cpp:132)."""
    N = Records(L, "con")
    con = N.zeros(batch, len(grids))
    for b in range(batch):
        rng = np.random.default_rng(BASE_SEED + 15485863 + first_instance + b)
        slack = np.abs(_rnd(rng, len(grids), L.dims.nc_max)) + 0.05
        dual = barrier / slack * (1.0 + 0.3 * _rnd(rng, len(grids), L.dims.nc_max))
        N.f(con[b], "slack")[...] = slack
        N.f(con[b], "dual")[...] = dual
        N.f(con[b], "residual")[...] = 0.01 * _rnd(rng, len(grids), L.dims.nc_max)
        N.f(con[b], "cmpl")[...] = slack * dual - barrier
    return con


def make_cone_batch(L, grids, batch, max_contacts, first_instance=0):
    """Friction-cone Jacobians of the active contacts of every grid point (RTOC_BUF_CONE record,
    include/rtoc_layout.h): dg_dq 5 x nv and dg_df 5 x 3 per contact, as FrictionCone::evalDerivatives
    would leave them (friction_cone.cpp:143-191).  dg_df is the rotated cone matrix
    [[0,0,-1],[1,0,-mu'],[-1,0,-mu'],[0,1,-mu'],[0,-1,-mu']] R^T, dg_dq a dense random block."""
    from .types import cone_dgdf_off, cone_stride
    nv = L.dims.nv
    cs, off = cone_stride(nv, max_contacts), cone_dgdf_off(nv, max_contacts)
    out = np.zeros((batch, len(grids), cs))
    mu = 0.7 / np.sqrt(2.0)
    cone = np.array([[0, 0, -1.0], [1, 0, -mu], [-1, 0, -mu], [0, 1, -mu], [0, -1, -mu]])
    for b in range(batch):
        rng = np.random.default_rng(BASE_SEED + 7919 + first_instance + b)
        for i in range(len(grids)):
            for k in range(max_contacts):
                dq = 0.3 * rng.uniform(-1, 1, (5, nv))
                a = rng.uniform(-0.3, 0.3, 3)
                Rm = np.eye(3) + np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                df = cone @ Rm.T
                out[b, i, k * 5 * nv:(k + 1) * 5 * nv] = dq.T.reshape(-1)       # column-major, ld 5
                out[b, i, off + k * 15:off + (k + 1) * 15] = df.T.reshape(-1)
    return out


def make_wrench_cone_batch(L, grids, batch, max_contacts, cones):
    """RTOC_BUF_CONE record in the wrench layout (include/rtoc_layout.h): the 17 x 6 cone matrix of the
    k-th ACTIVE surface contact of every grid point at k*102 (column-major, ld 17) -- what
    ContactWrenchCone::updateCone leaves in ConstraintComponentData::J (contact_wrench_cone.cpp:306-313).
    cones: one 17 x 6 matrix per contact slot (e.g. capi.wrench_cone_matrix(X, Y, mu))."""
    from .types import wrench_cone_stride
    out = np.zeros((batch, len(grids), wrench_cone_stride(max_contacts)))
    for i, g in enumerate(grids):
        for k in range(min(g.dimf // 6, max_contacts)):
            out[:, i, k * 102:(k + 1) * 102] = np.asarray(cones[k]).T.reshape(-1)
    return out
