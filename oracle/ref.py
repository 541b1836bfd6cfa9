"""ctypes loader for oracle/_ref/librtoc_ref.so: the REFERENCE'S OWN sources (src/riccati, src/dynamics, src/core
of /root/reference) compiled by oracle/Makefile.ref against oracle/ref_shim (Eigen and Pinocchio are absent from
the image; see oracle/ref_shim/README.md).  TEST INFRASTRUCTURE ONLY: it pins the C restatement (oracle/*.c) and
generates the golden vectors under tests/golden/ (tests/golden/make_ref_golden.py).  The library can only be
(re)built where /root/reference exists; a prebuilt copy travels with the repository snapshot to the GPU box."""
import ctypes as C
import os
import subprocess

import numpy as np

from robotoc_amd.types import Grid, Layout, grid_array

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "librtoc_ref.so")
REFERENCE = os.environ.get("RTOC_REFERENCE", "/root/reference")
_LIB = None


def available():
    return os.path.exists(_SO) or os.path.isdir(os.path.join(REFERENCE, "src", "riccati"))


def build(force=False):
    """Compiles the reference sources where they lie (never copied); needs REFERENCE to exist."""
    if os.path.isdir(os.path.join(REFERENCE, "src", "riccati")):
        args = ["make", "-C", _HERE, "-f", "Makefile.ref", "-j8", "REF=" + REFERENCE] + (["-B"] if force else [])
        subprocess.check_call(args, stdout=subprocess.DEVNULL)
    if not os.path.exists(_SO):
        raise RuntimeError("oracle/_ref/librtoc_ref.so is not built and %s does not exist" % REFERENCE)
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        dp, LP, GP = C.POINTER(C.c_double), C.POINTER(Layout), C.POINTER(Grid)
        _LIB.ref_riccati_sweep.argtypes = [LP, GP, C.c_int, dp, dp, dp, C.c_double, C.c_int, C.c_int]
        _LIB.ref_unconstr_sweep.argtypes = [LP, C.c_int, C.c_double, dp, dp, dp, C.c_int]
        _LIB.ref_condense_stage.argtypes = [LP, GP, dp, dp, C.c_double, C.c_int]
        _LIB.ref_expand_stage.argtypes = [LP, GP, dp, dp, dp, C.c_int]
        _LIB.ref_correct_costate.argtypes = [LP, dp, dp]
    return _LIB


def unconstr_update_solution(nv, N, dt, cost, limits, barrier, tau, x0, sol, rnea, con=None, init_constraints=True):
    """UnconstrOCPSolver::updateSolution with the reference's own stage / cost / constraints / dynamics / Riccati sources and
    injected inverse dynamics (oracle/ref_shim/ref_unconstr_solver_capi.cpp).  sol [N+1, 7 nv] is updated in place; returns
    (condensed KKT blocks per grid point, KKT error (sum of squares), primal step, dual step, con)."""
    L = lib()
    dp = C.POINTER(C.c_double)
    L.ref_unconstr_update_solution.argtypes = [C.c_int, C.c_int, C.c_double, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, C.c_int, dp, dp]
    nx = 2 * nv
    per = nx * nx + nx * nv + nv * nv + nx + nv + nx
    kkt = np.zeros((N + 1, per))
    out = np.zeros(3)
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    lim = None if limits is None else np.ascontiguousarray(limits, dtype=np.float64)
    if con is None:
        con = np.zeros((N, 2, 6 * nv))
    q0, v0 = np.ascontiguousarray(x0[:nv]), np.ascontiguousarray(x0[nv:])
    rc = L.ref_unconstr_update_solution(nv, N, dt, _p(cost), _p(lim) if lim is not None else None, barrier, tau, _p(q0), _p(v0), _p(sol),
                                        _p(rnea), _p(con), int(init_constraints), _p(kkt), _p(out))
    assert rc == 0
    return kkt, out[0], out[1], out[2], con


def unconstr_line_search_iteration(nv, N, dt, cost, limits, barrier, tau, x0, sol, rnea, inverse_dynamics, settings, con=None,
                                   init_constraints=True, max_trials=24):
    """UnconstrOCPSolver::updateSolution WITH enable_line_search (unconstr_ocp_solver.cpp:96-118): the reference's own
    UnconstrDirectMultipleShooting, UnconstrRiccatiRecursion and UnconstrLineSearch (oracle/ref_shim/ref_unconstr_ls_capi.cpp).
    inverse_dynamics(q, v, a) -> ID supplies what the trial iterates' stages ask of Pinocchio; settings = (step_size_reduction_rate,
    min_step_size, filter_cost_reduction_rate, filter_constraint_violation_reduction_rate).  sol [N+1, 7 nv] is updated in place.
    Returns dict(direction [N+1, 4 nv], dslack, kkt_error (sum of squares), max_primal, max_dual, eval (cost, barrier, violation),
    step, trials (filter evaluations consumed), con)."""
    L = lib()
    dp = C.POINTER(C.c_double)
    L.ref_uls_direction.argtypes = [C.c_int, C.c_int, C.c_double, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, C.c_int, dp, dp, dp]
    L.ref_uls_line_search.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp]
    L.ref_uls_integrate.argtypes = [dp, dp]
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    lim = None if limits is None else np.ascontiguousarray(limits, dtype=np.float64)
    if con is None:
        con = np.zeros((N, 2, 6 * nv))
    q0, v0 = np.ascontiguousarray(x0[:nv]), np.ascontiguousarray(x0[nv:])
    d = np.zeros((N + 1, 4 * nv))
    dslack = np.zeros((N, 6 * nv))
    out = np.zeros(6)
    rc = L.ref_uls_direction(nv, N, dt, _p(cost), _p(lim) if lim is not None else None, barrier, tau, _p(q0), _p(v0), _p(sol), _p(rnea),
                             _p(con), int(init_constraints), _p(d), _p(dslack), _p(out))
    assert rc == 0
    rate, min_step, cost_rate, viol_rate = settings
    trial = np.zeros((max_trials, N, nv))
    alpha = out[1]
    for k in range(max_trials):   # the loop of unconstr_line_search.cpp:51-64 visits max, max rate, max rate^2, ... until accepted or <= min
        for i in range(N):
            trial[k, i] = inverse_dynamics(sol[i, :nv] + alpha * d[i, :nv], sol[i, nv:2 * nv] + alpha * d[i, nv:2 * nv],
                                           sol[i, 2 * nv:3 * nv] + alpha * d[i, 2 * nv:3 * nv])
        alpha *= rate
    ls = np.zeros(2)
    rc = L.ref_uls_line_search(_p(trial), max_trials, rate, min_step, cost_rate, viol_rate, _p(ls))
    assert rc == 0
    used = max_trials - int(round(ls[1]))
    assert 0 < used < max_trials, "the reference asked for more trial iterates than were prepared"
    rc = L.ref_uls_integrate(_p(sol), _p(con))
    assert rc == 0
    return dict(direction=d, dslack=dslack, kkt_error=out[0], max_primal=out[1], max_dual=out[2], eval=out[3:6].copy(), step=ls[0],
                trials=used, con=con)


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def riccati_sweep(L, grids, kkt, ric, dirs, max_dts0=0.1, contact_dim=3, forward=True):
    """RiccatiRecursion::backward (+ forward) RiccatiRecursion of ONE instance ([stages, stride] arrays); kkt is
    mutated in place like the reference's containers, dirs[0].dx is the input."""
    lib().ref_riccati_sweep(C.byref(L), grid_array(grids), len(grids), _p(kkt), _p(ric), _p(dirs), max_dts0,
                            contact_dim, int(forward))


def riccati_sweep_bench(L, grids, kkt, dx0, reps, max_dts0=0.1, contact_dim=3):
    """`reps` timed backward + forward recursions of ONE instance ([stages, stride] record, read-only) by the reference's own
    RiccatiRecursion, one thread.  Returns dict(seconds, reload_seconds, sweeps).  What this build of the reference sources is:
    see ref_capi.cpp: ref_riccati_sweep_bench."""
    f = lib().ref_riccati_sweep_bench
    f.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int, C.c_int,
                  C.POINTER(C.c_double)]
    out = np.zeros(2)
    rc = f(C.byref(L), grid_array(grids), len(grids), _p(kkt), _p(dx0), max_dts0, contact_dim, reps, _p(out))
    assert rc == 0
    return dict(seconds=out[0], reload_seconds=out[1], sweeps=reps)


def unconstr_sweep(L, nstages, dt, kkt, ric, dirs, forward=True):
    lib().ref_unconstr_sweep(C.byref(L), nstages, dt, _p(kkt), _p(ric), _p(dirs), int(forward))


def condense_stage(L, g, kkt_rec, cdd_rec, damping=0.0, contact_dim=3):
    """condenseContactDynamics / condenseImpactDynamics of one grid point (no evalKKT-tail scalings)."""
    lib().ref_condense_stage(C.byref(L), C.byref(g), _p(kkt_rec), _p(cdd_rec), damping, contact_dim)


def expand_stage(L, g, cdd_rec, dir_rec, dir_next_rec, contact_dim=3):
    lib().ref_expand_stage(C.byref(L), C.byref(g), _p(cdd_rec), _p(dir_rec), _p(dir_next_rec), contact_dim)


def split_solution_integrate(L, g, step, dir_rec, sol_rec, q_integrated=None, contact_dim=3):
    """SplitSolution::integrate (src/core/split_solution.cpp:58-90) by the reference's own source on one record pair; sol_rec is
    updated in place.  q_integrated: robot.integrateConfiguration's result for a floating base (injected)."""
    fn = lib().ref_split_solution_integrate
    fn.argtypes = [C.POINTER(Layout), C.POINTER(Grid), C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    assert fn(C.byref(L), C.byref(g), contact_dim, step, _p(dir_rec), _p(sol_rec), _p(q_integrated) if q_integrated is not None else None) == 0


def sto_eval_kkt(grids, t, h, qtt, min_dwell, barrier=1.0e-3, fraction=0.995, sto_reg=0.0, cost_w=None, cost_tref=None):
    """SwitchingTimeOptimization::evalKKT (src/sto/switching_time_optimization.cpp:79-137) by the reference's own sources, with
    its STOCostFunction and minimum-dwell-time STOConstraints (oracle/ref_shim/ref_sto_capi.cpp).  h / qtt: [stages] the per-grid
    SplitKKTResidual::h / SplitKKTMatrix::Qtt, updated in place.  Returns lt_, diag(Qtt_) as the reference scattered them and
    [STO kkt_error, the dwell-time constraints' KKTError, dual feasibility, cost]."""
    fn = lib().ref_sto_eval_kkt
    dp = C.POINTER(C.c_double)
    fn.argtypes = [C.POINTER(Grid), dp, C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp]
    fn.restype = C.c_int
    n = len(grids)
    nev = sum(1 for g in grids[:-1] if g.type in (1, 2))
    t = np.ascontiguousarray(t, dtype=np.float64)
    min_dwell = np.ascontiguousarray(min_dwell, dtype=np.float64)
    assert t.shape == (n,) and h.shape == (n,) and qtt.shape == (n,) and min_dwell.shape == (nev + 1,)
    lt, qd, perf = np.zeros(max(nev, 1)), np.zeros(max(nev, 1)), np.zeros(4)
    w = np.ascontiguousarray(cost_w, dtype=np.float64) if cost_w is not None else None
    tr = np.ascontiguousarray(cost_tref, dtype=np.float64) if cost_tref is not None else None
    got = fn(grid_array(grids), _p(t), n, _p(h), _p(qtt), _p(min_dwell), barrier, fraction, sto_reg, _p(w) if w is not None else None,
             _p(tr) if tr is not None else None, _p(lt), _p(qd), _p(perf))
    assert got == nev, (got, nev)
    return lt[:nev], qd[:nev], perf


def correct_costate(L, se3_rec, dir_rec):
    lib().ref_correct_costate(C.byref(L), _p(se3_rec), _p(dir_rec))


# ---- the reference's own TimeDiscretization and LineSearchFilter (oracle/_ref/librtoc_ref_td.so) ----
_LIB_TD = None


def lib_td():
    global _LIB_TD
    if _LIB_TD is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "_ref", "librtoc_ref_td.so"))
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.ref_td_discretize.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, ip, dp, ip, C.c_int, ip, dp, dp]
        L.ref_td_discretize.restype = C.c_int
        L.ref_filter_create.argtypes = [C.c_double, C.c_double]
        L.ref_filter_create.restype = C.c_void_p
        L.ref_filter_destroy.argtypes = [C.c_void_p]
        L.ref_filter_clear.argtypes = [C.c_void_p]
        L.ref_filter_try.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ref_filter_try.restype = C.c_int
        _LIB_TD = L
    return _LIB_TD


def discretize(T, N, t, events, phase_based):
    """TimeDiscretization::discretize (+ correctTimeSteps): events = [(kind 'impact'|'lift', time, sto)].
    Returns (table [n+1, 11] ints: type, phase, sto, sto_next, switching_constraint, stage_in_phase, num_grids_in_phase,
    impact_index, lift_index, stage, 0; dt [n+1]; t [n+1]; maxTimeStep)."""
    ne = len(events)
    kind = np.array([0 if e[0] == "impact" else 1 for e in events], dtype=np.int32)
    time = np.array([e[1] for e in events], dtype=np.float64)
    sto = np.array([int(e[2]) for e in events], dtype=np.int32)
    cap = N + 2 + 3 * ne
    oi, od, mx = np.zeros((cap, 11), dtype=np.int32), np.zeros((cap, 2)), C.c_double()
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    n = lib_td().ref_td_discretize(T, N, t, ne, kind.ctypes.data_as(ip), time.ctypes.data_as(dp), sto.ctypes.data_as(ip),
                                   int(phase_based), oi.ctypes.data_as(ip), od.ctypes.data_as(dp), C.byref(mx))
    return oi[:n + 1], od[:n + 1, 0].copy(), od[:n + 1, 1].copy(), mx.value


class LineSearchFilter:
    def __init__(self, cost_reduction_rate=0.005, constraint_violation_reduction_rate=0.005):
        self._h = lib_td().ref_filter_create(cost_reduction_rate, constraint_violation_reduction_rate)

    def try_step(self, cost, violation):
        return lib_td().ref_filter_try(self._h, cost, violation)

    def clear(self):
        lib_td().ref_filter_clear(self._h)

    def __del__(self):
        if self._h:
            lib_td().ref_filter_destroy(self._h)
            self._h = None
