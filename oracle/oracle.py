"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It never falls back to anything: if librtoc_oracle.so cannot be
built/loaded an exception is raised.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from robotoc_amd.types import BoxRow, Dims, Grid, Layout, grid_array

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "librtoc_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("rtoc_oracle.c", "rtoc_oracle_condense.c", "rtoc_oracle_bench.c", "rtoc_oracle_rbd.c", "rtoc_oracle_rbd_cs.c",
                                              "rtoc_oracle_aba.c", "rtoc_oracle_aba_cs.c")]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("rtoc.h", "rtoc_layout.h", "rtoc_robot.h")]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librtoc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


_SRCS = ("rtoc_oracle.c", "rtoc_oracle_condense.c", "rtoc_oracle_bench.c", "rtoc_oracle_rbd.c", "rtoc_oracle_rbd_cs.c", "rtoc_oracle_aba.c",
         "rtoc_oracle_aba_cs.c")
_NATIVE = None


def _host_tag():
    """Names the host CPU (model + ISA flags): a -march=native object must never be loaded on another machine."""
    import hashlib
    model, flags = "", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            if line.startswith("flags") and not flags:
                flags = line.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    return hashlib.sha1((model + "|" + flags).encode()).hexdigest()[:12], model, ("avx512f" in flags.split())


def native_lib():
    """The same C restatement built -O3 -march=native ON THIS HOST (the CPU baseline's build), for bench.py's
    cpu_baseline leg only -- the parity suite keeps the portable x86-64-v3 object that travels with the snapshot.
    Returns (CDLL with the orc_bench_* entry points typed, description dict)."""
    global _NATIVE
    if _NATIVE is None:
        tag, model, avx512 = _host_tag()
        out_dir = os.path.join(_HERE, "_native")
        os.makedirs(out_dir, exist_ok=True)
        so = os.path.join(out_dir, "librtoc_oracle_%s.so" % tag)
        srcs = [os.path.join(_HERE, f) for f in _SRCS]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=c99", "-w", "-shared", "-o", so] + srcs + ["-lm"])
        L = C.CDLL(so)
        _type_bench(L)
        _NATIVE = (L, dict(flags="-O3 -march=native -fopenmp", host_cpu=model, avx512=bool(avx512)))
    return _NATIVE


def _type_bench(L):
    dp = C.POINTER(C.c_double)
    L.orc_bench_sweep.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, dp, dp, C.c_double, C.c_int, C.c_int, dp]
    L.orc_bench_sweep.restype = C.c_uint
    L.orc_bench_sqp.restype = C.c_uint
    L.orc_bench_sqp.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, dp, dp, dp, dp, dp, C.POINTER(BoxRow), C.c_int,
                                C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, dp]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        _LIB.orc_layout.argtypes = [C.POINTER(Dims), C.POINTER(Layout)]
        _LIB.orc_riccati_backward.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, dp, dp,
                                              C.c_double]
        _LIB.orc_riccati_backward.restype = C.c_uint
        _LIB.orc_riccati_forward.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, dp, dp,
                                             dp]
        _LIB.orc_unconstr_backward.argtypes = [C.POINTER(Layout), C.c_int, C.c_double, dp, dp]
        _LIB.orc_unconstr_backward.restype = C.c_uint
        _LIB.orc_unconstr_forward.argtypes = [C.POINTER(Layout), C.c_int, C.c_double, dp, dp, dp]
        _LIB.orc_riccati_sweep_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int,
                                                 C.c_int, dp, dp, dp, dp, C.c_double,
                                                 C.POINTER(C.c_uint), C.c_int, C.c_int]
        _LIB.orc_unconstr_sweep_batch.argtypes = [C.POINTER(Layout), C.c_int, C.c_int, C.c_double,
                                                  dp, dp, dp, dp, C.POINTER(C.c_uint)]
        _LIB.orc_stage_backward.argtypes = [C.POINTER(Layout), dp, dp, dp, C.c_int, C.c_int, C.c_int]
        _LIB.orc_stage_backward.restype = C.c_uint
        _LIB.orc_stage_backward_impact.argtypes = [C.POINTER(Layout), dp, dp, dp, C.c_int]
        _LIB.orc_stage_phase_transition.argtypes = [C.POINTER(Layout), dp, dp, dp, C.c_int,
                                                    C.c_double]
        _LIB.orc_compute_MJtJinv.argtypes = [C.c_int, C.c_int, dp, dp, C.c_int, C.c_double, dp, C.c_int]
        _LIB.orc_compute_MJtJinv.restype = C.c_int
        _LIB.orc_condense_stage.argtypes = [C.POINTER(Layout), C.POINTER(Grid), dp, dp, C.c_double]
        _LIB.orc_condense_stage.restype = C.c_uint
        _LIB.orc_condense_impact_stage.argtypes = [C.POINTER(Layout), C.POINTER(Grid), dp, dp, C.c_double]
        _LIB.orc_condense_impact_stage.restype = C.c_uint
        _LIB.orc_expand_stage.argtypes = [C.POINTER(Layout), C.POINTER(Grid), dp, dp, dp]
        _LIB.orc_condense_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, dp, dp,
                                            C.c_double, C.POINTER(C.c_uint)]
        _LIB.orc_expand_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, dp, dp]
        _LIB.orc_pdipm_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int,
                                         C.POINTER(BoxRow), C.c_int, dp, dp, dp, C.c_double, dp, C.c_int, dp]
        _LIB.orc_state_correction_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int,
                                                    dp, dp, dp, dp]
        _LIB.orc_unconstr_dynamics_batch.argtypes = [C.POINTER(Layout), C.c_int, C.c_int, dp, dp, dp,
                                                     C.c_double, C.c_int]
        _LIB.orc_cone_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, C.c_int, C.c_int,
                                        dp, dp, dp, dp, dp, C.c_double, dp, C.c_int]
        _LIB.orc_kkt_error.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, dp, dp, dp, C.POINTER(BoxRow),
                                       C.c_int, C.c_int, C.c_int, C.c_int]
        _LIB.orc_kkt_error.restype = C.c_double
        _LIB.orc_wrench_cone_matrix.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        _LIB.orc_wrench_cone_matrix.restype = None
        _LIB.orc_wrench_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, C.c_int, dp, dp, dp,
                                          dp, C.c_double, dp, C.c_int]
        _LIB.orc_wrench_batch.restype = None
        _LIB.orc_integrate_solution_batch.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, dp, dp, dp]
        _LIB.orc_sto_eval_kkt.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, dp, dp, dp, C.c_int]
        _LIB.orc_sto_eval_kkt.restype = C.c_double
        _LIB.orc_bench_sweep.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, dp, dp, C.c_double,
                                         C.c_int, C.c_int, dp]
        _LIB.orc_bench_sweep.restype = C.c_uint
        _LIB.orc_bench_sqp.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_int, dp, dp, dp, dp, dp,
                                       C.POINTER(BoxRow), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_int, C.c_int, dp]
        _LIB.orc_bench_sqp.restype = C.c_uint
    return _LIB


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def layout(dims):
    L = Layout()
    lib().orc_layout(C.byref(dims), C.byref(L))
    return L


def riccati_backward(L, grids, kkt, ric, max_dts0=0.1):
    """kkt, ric: [stages, stride] arrays of ONE instance (kkt is mutated in place like the reference)."""
    g = grid_array(grids)
    return lib().orc_riccati_backward(C.byref(L), g, len(grids), _p(kkt), _p(ric), max_dts0)


def riccati_forward(L, grids, kkt, ric, dirs):
    g = grid_array(grids)
    lib().orc_riccati_forward(C.byref(L), g, len(grids), _p(kkt), _p(ric), _p(dirs))


def unconstr_backward(L, nstages, dt, kkt, ric):
    return lib().orc_unconstr_backward(C.byref(L), nstages, dt, _p(kkt), _p(ric))


def unconstr_forward(L, nstages, dt, kkt, ric, dirs):
    lib().orc_unconstr_forward(C.byref(L), nstages, dt, _p(kkt), _p(ric), _p(dirs))


def riccati_sweep_batch(L, grids, kkt, ric, dirs, dx0=None, max_dts0=0.1, backward=True,
                        forward=True):
    """Batched (OpenMP over instances) sweep; arrays are [batch, stages, stride]."""
    g = grid_array(grids)
    batch = kkt.shape[0]
    stat = (C.c_uint * batch)()
    lib().orc_riccati_sweep_batch(C.byref(L), g, len(grids), batch, _p(kkt), _p(ric), _p(dirs),
                                  _p(dx0) if dx0 is not None else None, max_dts0, stat,
                                  int(backward), int(forward))
    return np.frombuffer(stat, dtype=np.uint32).copy()


def unconstr_sweep_batch(L, nstages, dt, kkt, ric, dirs, dx0=None):
    batch = kkt.shape[0]
    stat = (C.c_uint * batch)()
    lib().orc_unconstr_sweep_batch(C.byref(L), nstages, batch, dt, _p(kkt), _p(ric), _p(dirs),
                                   _p(dx0) if dx0 is not None else None, stat)
    return np.frombuffer(stat, dtype=np.uint32).copy()


def stage_backward(L, kkt_rec, ric_next_rec, ric_out_rec, ns=0, sto=False, sto_next=False):
    """RiccatiFactorizer::backwardRiccatiRecursion (7-arg) on single records."""
    return lib().orc_stage_backward(C.byref(L), _p(kkt_rec), _p(ric_next_rec), _p(ric_out_rec), ns,
                                    int(sto), int(sto_next))


def stage_backward_impact(L, kkt_rec, ric_next_rec, ric_out_rec, sto=False):
    lib().orc_stage_backward_impact(C.byref(L), _p(kkt_rec), _p(ric_next_rec), _p(ric_out_rec),
                                    int(sto))


def stage_phase_transition(L, ric_rec, ric_m_rec, policy_rec, sto_next, max_dts0):
    lib().orc_stage_phase_transition(C.byref(L), _p(ric_rec), _p(ric_m_rec), _p(policy_rec),
                                     int(sto_next), max_dts0)


def compute_MJtJinv(M, J, damping=0.0):
    """Robot::computeMJtJinv on numpy arrays (M nv x nv, J nf x nv)."""
    nv, nf = M.shape[0], J.shape[0]
    Mf = np.asfortranarray(M).ravel(order="K").copy()
    Jf = np.asfortranarray(J).ravel(order="K").copy() if nf else np.zeros(1)
    out = np.zeros((nv + nf) * (nv + nf))
    bad = lib().orc_compute_MJtJinv(nv, nf, _p(Mf), _p(Jf), max(nf, 1), damping, _p(out), nv + nf)
    return out.reshape(nv + nf, nv + nf).T.copy(), bad


def condense_stage(L, g, kkt_rec, cdd_rec, damping=0.0):
    from robotoc_amd.types import GRID_IMPACT
    f = lib().orc_condense_impact_stage if g.type == GRID_IMPACT else lib().orc_condense_stage
    return f(C.byref(L), C.byref(g), _p(kkt_rec), _p(cdd_rec), damping)


def expand_stage(L, g, cdd_rec, dir_rec, dir_next_rec):
    lib().orc_expand_stage(C.byref(L), C.byref(g), _p(cdd_rec), _p(dir_rec), _p(dir_next_rec))


def condense_batch(L, grids, kkt, cdd, damping=0.0):
    g = grid_array(grids)
    batch = kkt.shape[0]
    stat = (C.c_uint * batch)()
    lib().orc_condense_batch(C.byref(L), g, len(grids), batch, _p(kkt), _p(cdd), damping, stat)
    return np.frombuffer(stat, dtype=np.uint32).copy()


def expand_batch(L, grids, cdd, dirs):
    g = grid_array(grids)
    lib().orc_expand_batch(C.byref(L), g, len(grids), cdd.shape[0], _p(cdd), _p(dirs))


def _rows(rows):
    arr = (BoxRow * max(len(rows), 1))()
    for i, r in enumerate(rows):
        arr[i] = r
    return arr


def pdipm_condense_batch(L, grids, rows, kkt, con, cdd=None):
    """cdd: the ContactDynamicsData records (Qaa diagonal, la) the acceleration-limit rows (VAR_A) act on"""
    assert cdd is not None or not any(r.var == 3 for r in rows)
    lib().orc_pdipm_batch(C.byref(L), grid_array(grids), len(grids), kkt.shape[0], _rows(rows), len(rows),
                          _p(kkt), _p(con), None, 0.0, None, 0, _p(cdd) if cdd is not None else None)


def pdipm_expand_batch(L, grids, rows, con, dirs, tau):
    steps = np.ones((con.shape[0], 2))
    lib().orc_pdipm_batch(C.byref(L), grid_array(grids), len(grids), con.shape[0], _rows(rows), len(rows),
                          None, _p(con), _p(dirs), tau, _p(steps), 1, None)
    return steps


def pdipm_update_batch(L, grids, rows, con, steps):
    steps = np.ascontiguousarray(steps, dtype=np.float64)
    lib().orc_pdipm_batch(C.byref(L), grid_array(grids), len(grids), con.shape[0], _rows(rows), len(rows),
                          None, _p(con), None, 0.0, _p(steps), 2, None)


def state_correction_batch(L, grids, se3, kkt=None, dirs=None, dx0=None):
    """correctLinearize(Impact)StateEquation on kkt, correctCostateDirection on dirs,
    computeInitialStateDirection on dx0 (each optional), state_equation.cpp:68-109."""
    lib().orc_state_correction_batch(C.byref(L), grid_array(grids), len(grids), se3.shape[0], _p(se3),
                                     _p(kkt) if kkt is not None else None,
                                     _p(dirs) if dirs is not None else None,
                                     _p(dx0) if dx0 is not None else None)


def unconstr_condense_batch(L, nstages, kkt, cdd):
    """UnconstrDynamics::condenseUnconstrDynamics on every non-terminal grid point."""
    lib().orc_unconstr_dynamics_batch(C.byref(L), nstages, kkt.shape[0], _p(kkt), _p(cdd), None, 1.0, 0)


def unconstr_expand_batch(L, nstages, cdd, dirs, dt):
    """UnconstrDynamics::expandPrimal + expandDual on every non-terminal grid point."""
    lib().orc_unconstr_dynamics_batch(C.byref(L), nstages, cdd.shape[0], None, _p(cdd), _p(dirs), dt, 1)


def cone_condense_batch(L, grids, max_contacts, contact_dim, cone, kkt, cdd, con):
    """FrictionCone::condenseSlackAndDual on every non-terminal grid point (friction_cone.cpp:194-235)."""
    lib().orc_cone_batch(C.byref(L), grid_array(grids), len(grids), kkt.shape[0], max_contacts, contact_dim,
                         _p(cone), _p(kkt), _p(cdd), _p(con), None, 0.0, None, 0)


def cone_expand_batch(L, grids, max_contacts, contact_dim, cone, con, dirs, tau, steps):
    """FrictionCone::expandSlackAndDual + step sizes; `steps` [batch,2] is min-reduced in place."""
    lib().orc_cone_batch(C.byref(L), grid_array(grids), len(grids), con.shape[0], max_contacts, contact_dim,
                         _p(cone), None, None, _p(con), _p(dirs), tau, _p(steps), 1)


def cone_update_batch(L, grids, max_contacts, contact_dim, con, steps):
    steps = np.ascontiguousarray(steps, dtype=np.float64)
    lib().orc_cone_batch(C.byref(L), grid_array(grids), len(grids), con.shape[0], max_contacts, contact_dim,
                         None, None, None, _p(con), None, 0.0, _p(steps), 2)


def wrench_cone_matrix(X, Y, mu):
    """ContactWrenchCone::computeCone (contact_wrench_cone.cpp:282-303) -> 17 x 6 array."""
    out = np.zeros(102)
    lib().orc_wrench_cone_matrix(X, Y, mu, _p(out))
    return out.reshape(6, 17).T.copy()


def wrench_condense_batch(L, grids, max_contacts, cone, cdd, con):
    """ContactWrenchCone::condenseSlackAndDual on every non-terminal grid point (:209-238)."""
    _wrench(L, grids, cdd.shape[0], max_contacts, cone, cdd, con, None, 0.0, None, 0)


def wrench_expand_batch(L, grids, max_contacts, cone, con, dirs, tau, steps):
    """ContactWrenchCone::expandSlackAndDual + step sizes; `steps` [batch,2] is min-reduced in place."""
    _wrench(L, grids, con.shape[0], max_contacts, cone, None, con, dirs, tau, steps, 1)


def wrench_update_batch(L, grids, max_contacts, con, steps):
    _wrench(L, grids, con.shape[0], max_contacts, None, None, con, None, 0.0,
            np.ascontiguousarray(steps, dtype=np.float64), 2)


def _wrench(L, grids, batch, max_contacts, cone, cdd, con, dirs, tau, steps, phase):
    opt = lambda a: _p(a) if a is not None else None
    lib().orc_wrench_batch(C.byref(L), grid_array(grids), len(grids), batch, max_contacts, opt(cone), opt(cdd),
                           opt(con), opt(dirs), tau, opt(steps), phase)


def kkt_error(L, grids, kkt, cdd=None, con=None, rows=(), cone_contacts=0, cone_dim=3, cone_rows=5):
    """OCPSolver::KKTError() (without the STO term) of every instance, pre-condensation records."""
    out = np.zeros(kkt.shape[0])
    for b in range(kkt.shape[0]):
        out[b] = lib().orc_kkt_error(C.byref(L), grid_array(grids), len(grids), _p(kkt[b]),
                                     _p(cdd[b]) if cdd is not None else None,
                                     _p(con[b]) if con is not None else None, _rows(rows), len(rows),
                                     cone_contacts, cone_dim, cone_rows)
    return out


def integrate_solution_batch(L, grids, steps, dirs, sol):
    """SplitSolution::integrate (Euclidean members) with the primal step sizes steps[:, 0]."""
    steps = np.ascontiguousarray(steps, dtype=np.float64)
    lib().orc_integrate_solution_batch(C.byref(L), grid_array(grids), len(grids), sol.shape[0], _p(steps), _p(dirs),
                                       _p(sol))


def bench_sweep(L, grids, kkt, dx0, reps, nthreads=0, max_dts0=0.1, native=False):
    """Timed backward+forward sweeps of reps x batch instances, thread-private working records
    (rtoc_oracle_bench.c).  The inputs are read-only.  Returns dict(seconds, refill_seconds, threads, sweeps).
    native: the -march=native build of this host (native_lib)."""
    out = np.zeros(4)
    st = (native_lib()[0] if native else lib()).orc_bench_sweep(C.byref(L), grid_array(grids), len(grids), kkt.shape[0], _p(kkt), _p(dx0), max_dts0,
                               reps, nthreads, _p(out))
    return dict(seconds=out[0], refill_seconds=out[1], threads=int(out[2]), sweeps=reps * kkt.shape[0], status=st)


def bench_sqp(L, grids, kkt, cdd, con, cone, dx0, rows, max_contacts, contact_dim, tau, reps, nthreads=0,
              max_dts0=0.1, native=False):
    """Timed SQP hot-path iterations (condense -> sweep -> expand -> step sizes -> update) of reps x batch instances
    on pre-condensation records, thread-private working records.  Inputs are read-only."""
    out = np.zeros(4)
    st = (native_lib()[0] if native else lib()).orc_bench_sqp(C.byref(L), grid_array(grids), len(grids), kkt.shape[0], _p(kkt), _p(cdd), _p(con),
                             _p(cone) if cone is not None else None, _p(dx0), _rows(rows), len(rows), max_contacts,
                             contact_dim, tau, max_dts0, reps, nthreads, _p(out))
    return dict(seconds=out[0], refill_seconds=out[1], threads=int(out[2]), iterations=reps * kkt.shape[0], status=st)


def sto_eval_kkt(L, grids, kkt, lt, qtt_diag):
    """SwitchingTimeOptimization::evalKKT scatter + STO KKT-error term (switching_time_optimization.cpp:105-137) of
    every instance: kkt [batch, stages, stride] (condensed records, mutated), lt / qtt_diag [batch, num_events].
    Returns the squared STO KKT error per instance."""
    lt = np.ascontiguousarray(lt, dtype=np.float64)
    qtt_diag = np.ascontiguousarray(qtt_diag, dtype=np.float64)
    out = np.zeros(kkt.shape[0])
    for b in range(kkt.shape[0]):
        out[b] = lib().orc_sto_eval_kkt(C.byref(L), grid_array(grids), len(grids), _p(kkt[b]), _p(lt[b]), _p(qtt_diag[b]),
                                        lt.shape[1])
    return out


# ---- rigid-body side (rtoc_oracle_rbd.c; PARITY UNPINNED, see its header) ----
def _rbd():
    from robotoc_amd.robot_model import RobotModel
    L = lib()
    if not getattr(L, "_rbd_ready", False):
        dp, mp = C.POINTER(C.c_double), C.POINTER(RobotModel)
        L.orc_rbd_integrate.argtypes = [mp, dp, dp, C.c_double, dp]
        L.orc_rbd_eval.argtypes = [mp, C.c_int, dp, dp, dp, dp, dp, C.c_uint, dp, dp]
        L.orc_rbd_eval.restype = C.c_int
        L.orc_rbd_eval_ex.argtypes = [mp, C.c_int, dp, dp, dp, dp, dp, C.c_uint, dp, dp, dp]
        L.orc_rbd_eval_ex.restype = C.c_int
        L.orc_rbd_linearize_fd_ex.argtypes = [mp, C.c_int, dp, dp, dp, dp, dp, C.c_uint, dp, dp, C.c_double, dp, dp, dp, C.c_int]
        L.orc_rbd_linearize_cs.argtypes = [mp, C.c_int, dp, dp, dp, dp, C.c_int, dp, C.c_int, C.c_uint, dp, dp, dp, dp, dp, C.c_int]
        L.orc_rbd_log6.argtypes = [dp, dp, dp]
        L.orc_se3_integrate.argtypes = [dp, dp, C.c_double, dp]
        L.orc_se3_difference.argtypes = [dp, dp, dp]
        L.orc_rbd_exp6.argtypes = [dp, dp, dp]
        L.orc_rbd_linearize_fd.argtypes = [mp, C.c_int, dp, dp, dp, dp, dp, C.c_uint, dp, C.c_double, dp, dp, dp, C.c_int]
        L.orc_rbd_mass_matrix_world.argtypes = [mp, dp, dp]
        L.orc_rbd_energy.argtypes = [mp, dp, dp, dp]
        L.orc_rbd_energy.restype = C.c_double
        L.orc_rbd_momentum_world.argtypes = [mp, dp, dp, dp]
        L.orc_rbd_contact_position.argtypes = [mp, dp, C.c_int, dp]
        L.orc_rbd_contact_placement.argtypes = [mp, dp, C.c_int, dp, dp]
        L.orc_aba_forward_dynamics.argtypes = [mp, dp, dp, dp, dp, C.c_uint, dp]
        L.orc_aba_crba.argtypes = [mp, dp, dp]
        L.orc_aba_linearize_cs.argtypes = [mp, dp, dp, dp, dp, C.c_int, C.c_uint, dp, dp]
        L._rbd_ready = True
    return L


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def rbd_integrate(model, q, dq, scale=1.0):
    q, dq = _c(q), _c(dq)
    out = np.empty_like(q)
    _rbd().orc_rbd_integrate(C.byref(model), _d(q), _d(dq), scale, _d(out))
    return out


def rbd_eval(model, impact, q, v, a, fstack, u, active, pref, rref=None):
    """[ID; C] of evalContactDynamics / evalImpactDynamics; rref: desired rotations [ncontacts, 9] of surface contacts"""
    q, v, a, fstack, u, pref = _c(q), _c(v), _c(a), _c(fstack), _c(u), _c(pref)
    rr = None if rref is None else _c(rref)
    out = np.zeros(model.nv + 6 * model.ncontacts)
    n = _rbd().orc_rbd_eval_ex(C.byref(model), int(impact), _d(q), _d(v), _d(a), _d(fstack), _d(u), int(active), _d(pref),
                               _d(rr) if rr is not None else None, _d(out))
    return out[:model.nv + n]


def rbd_linearize_fd(model, impact, q, v, a, fstack, u, active, pref, eps=1e-6, rref=None):
    q, v, a, fstack, u, pref = _c(q), _c(v), _c(a), _c(fstack), _c(u), _c(pref)
    rr = None if rref is None else _c(rref)
    n = model.nv + model.active_rows(int(active))
    D = [np.zeros((model.nv, n)) for _ in range(3)]  # column-major (n x nv) seen from C
    _rbd().orc_rbd_linearize_fd_ex(C.byref(model), int(impact), _d(q), _d(v), _d(a), _d(fstack), _d(u), int(active), _d(pref),
                                   _d(rr) if rr is not None else None, eps, _d(D[0]), _d(D[1]), _d(D[2]), n)
    return tuple(d.T.copy() for d in D)


def rbd_linearize_cs(model, impact, q, v, a, fstack, u, active, pref, rref=None):
    """Complex-step derivatives of rbd_eval (rtoc_oracle_rbd_cs.c): exact to rounding, no step size.  Same return as rbd_linearize_fd."""
    q, v, a, fstack, u, pref = _c(q), _c(v), _c(a), _c(fstack), _c(u), _c(pref)
    rr = None if rref is None else _c(rref)
    n = model.nv + model.active_rows(int(active))
    D = [np.zeros((model.nv, n)) for _ in range(3)]  # column-major (n x nv) seen from C
    _rbd().orc_rbd_linearize_cs(C.byref(model), int(impact), _d(q), _d(v), _d(a), _d(fstack), int(fstack.size), _d(u), int(u.size),
                                int(active), _d(pref), _d(rr) if rr is not None else None, _d(D[0]), _d(D[1]), _d(D[2]), n)
    return tuple(d.T.copy() for d in D)


# ---- the second formulation (rtoc_oracle_aba.c: articulated-body algorithm + composite-rigid-body algorithm in world coordinates) ----
def aba_forward_dynamics(model, q, v, tau, fstack, active):
    """a = FD(q, v, tau, f_ext): pinocchio::aba restated in world coordinates; tau [nv] generalised forces, fstack / active as rbd_eval."""
    q, v, tau, fstack = _c(q), _c(v), _c(tau), _c(fstack)
    a = np.zeros(model.nv)
    _rbd().orc_aba_forward_dynamics(C.byref(model), _d(q), _d(v), _d(tau), _d(fstack), int(active), _d(a))
    return a


def aba_crba(model, q):
    q = _c(q)
    M = np.zeros((model.nv, model.nv))
    _rbd().orc_aba_crba(C.byref(model), _d(q), _d(M))
    return M.T.copy()


def aba_linearize_cs(model, q, v, tau, fstack, active):
    """(dFD/dq, dFD/dv) [nv x nv] of aba_forward_dynamics by the complex step (rtoc_oracle_aba_cs.c), q on the manifold."""
    q, v, tau, fstack = _c(q), _c(v), _c(tau), _c(fstack)
    D = [np.zeros((model.nv, model.nv)) for _ in range(2)]
    _rbd().orc_aba_linearize_cs(C.byref(model), _d(q), _d(v), _d(tau), _d(fstack), int(fstack.size), int(active), _d(D[0]), _d(D[1]))
    return D[0].T.copy(), D[1].T.copy()


def rbd_log6(R, p):
    R, p = _c(R), _c(p)
    xi = np.zeros(6)
    _rbd().orc_rbd_log6(_d(R), _d(p), _d(xi))
    return xi


def rbd_exp6(xi):
    xi = _c(xi)
    R, p = np.zeros(9), np.zeros(3)
    _rbd().orc_rbd_exp6(_d(xi), _d(R), _d(p))
    return R.reshape(3, 3), p


def rbd_mass_matrix_world(model, q):
    q = _c(q)
    M = np.zeros((model.nv, model.nv))
    _rbd().orc_rbd_mass_matrix_world(C.byref(model), _d(q), _d(M))
    return M.T.copy()


def rbd_energy(model, q, v):
    q, v = _c(q), _c(v)
    U = C.c_double()
    T = _rbd().orc_rbd_energy(C.byref(model), _d(q), _d(v), C.byref(U))
    return T, U.value


def rbd_momentum_world(model, q, v):
    q, v = _c(q), _c(v)
    h = np.zeros(6)
    _rbd().orc_rbd_momentum_world(C.byref(model), _d(q), _d(v), _d(h))
    return h


def rbd_contact_position(model, q, c):
    q = _c(q)
    p = np.zeros(3)
    _rbd().orc_rbd_contact_position(C.byref(model), _d(q), int(c), _d(p))
    return p


def rbd_contact_placement(model, q, c):
    q = _c(q)
    R, p = np.zeros(9), np.zeros(3)
    _rbd().orc_rbd_contact_placement(C.byref(model), _d(q), int(c), _d(R), _d(p))
    return R.reshape(3, 3), p


def se3_integrate(q7, v6, scale=1.0):
    """pinocchio::integrate on a free-flyer: M exp6(scale v6)"""
    q7, v6 = _c(q7), _c(v6)
    out = np.zeros(7)
    _rbd().orc_se3_integrate(_d(q7), _d(v6), scale, _d(out))
    return out


def se3_difference(q0_7, qf_7):
    """pinocchio::difference on a free-flyer: log6(M0^-1 Mf)"""
    q0_7, qf_7 = _c(q0_7), _c(qf_7)
    out = np.zeros(6)
    _rbd().orc_se3_difference(_d(q0_7), _d(qf_7), _d(out))
    return out
