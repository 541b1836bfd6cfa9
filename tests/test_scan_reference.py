"""Horizon scan (RTOC_OPT_BACKWARD_SCAN), CPU side: (i) the numpy statement of the scan
(tests/scan_reference.py) reproduces the serial oracle's P, s on every grid point of the trot / jump
grids (lift, impact, switching-constraint grid points), (ii) the workgroup bodies the HIP kernels
instantiate (robotoc_amd/csrc/riccati_scan_core.hpp), compiled for the host with one thread per
workgroup, reproduce the numpy statement -- algebra and indexing of the kernels checked without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import rel_err
from robotoc_amd import grid as G, problems as pr
from robotoc_amd.types import GRID_IMPACT, Records, grid_array
from scan_reference import scan_backward

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# scan vs serial recursion: SURVEY 8c states <= 1e-8 for the scan variant
TOL_SCAN = 1e-8
# emulated kernel bodies vs numpy scan (same formulas, different summation order / pivoting)
TOL_EMU = 1e-9


def _emu():
    src = os.path.join(ROOT, "tests", "cpp", "scan_emulation.cpp")
    so = os.path.join(ROOT, "tests", "cpp", "libscan_emulation.so")
    core = os.path.join(ROOT, "robotoc_amd", "csrc", "riccati_scan_core.hpp")
    core2 = os.path.join(ROOT, "robotoc_amd", "csrc", "riccati_scan_sto.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core), os.path.getmtime(core2)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", src, "-o", so])
    lib = C.CDLL(so)
    lib.scan_emu_backward.restype = C.c_int
    lib.scan_emu_ps_stride.restype = C.c_int
    return lib


def _no_sto(grids):
    for g in grids:
        g.sto = 0
        g.sto_next = 0
    return grids


CASES = {
    "anymal_trot": lambda: pr.config_anymal_trot()[:2],
    "anymal_trot_short": lambda: pr.config_anymal_trot(N=12, dt=0.05)[:2],
    "anymal_jump": lambda: (lambda d, g, *_: (d, _no_sto(g)))(*pr.config_anymal_jump_sto()),
    "icub32_jump": lambda: (lambda d, g, *_: (d, _no_sto(g)))(*pr.config_icub_jump(nv=32)),
    "iiwa14_dense": lambda: pr.config_iiwa14()[:2],
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode", ["dynamics", "factory"])
def test_numpy_scan_matches_serial_oracle(oracle, case, mode):
    dims, grids = CASES[case]()
    L = oracle.layout(dims)
    kkt = pr.make_kkt_batch(L, grids, 1, mode=mode)[0]
    P, s, levels = scan_backward(L, grids, kkt)
    assert levels == int(np.ceil(np.log2(len(grids))))
    ric = Records(L, "ric").zeros(len(grids))
    assert oracle.riccati_backward(L, grids, kkt.copy(), ric) == 0
    R = Records(L, "ric")
    for i in range(len(grids)):
        assert rel_err(P[i], R.f(ric[i], "P")) <= TOL_SCAN, (i, "P")
        assert rel_err(s[i], R.f(ric[i], "s")) <= TOL_SCAN, (i, "s")


@pytest.mark.parametrize("case", sorted(CASES))
def test_kernel_bodies_emulated_on_host_match_numpy_scan(oracle, case):
    dims, grids = CASES[case]()
    L = oracle.layout(dims)
    n = len(grids)
    kkt = pr.make_kkt_batch(L, grids, 1, mode="dynamics")[0]
    lib = _emu()
    stride = lib.scan_emu_ps_stride(dims.nv)
    ps = np.zeros((n, stride))
    stat = C.c_uint(0)
    lv = lib.scan_emu_backward(dims.nv, dims.nu, dims.ns_max, grid_array(grids), n,
                               kkt.ctypes.data_as(C.POINTER(C.c_double)), ps.ctypes.data_as(C.POINTER(C.c_double)),
                               C.byref(stat))
    assert lv == int(np.ceil(np.log2(n))) and stat.value == 0
    P, s, _ = scan_backward(L, grids, kkt)
    nx = 2 * dims.nv
    for i in range(n):
        Pe = ps[i, :nx * nx].reshape(nx, nx).T
        se = ps[i, stride - ((nx + 7) & ~7):][:nx]
        assert rel_err(Pe, P[i]) <= TOL_EMU, (i, "P", rel_err(Pe, P[i]))
        assert rel_err(se, s[i]) <= TOL_EMU, (i, "s", rel_err(se, s[i]))


@pytest.mark.parametrize("case", sorted(CASES))
def test_forward_prefix_scan_bodies_match_serial_oracle(oracle, case):
    """The forward recursion as a prefix scan of the closed-loop maps (host emulation of the kernel bodies)
    against the oracle's serial forward recursion on the oracle's own factorisation."""
    from helpers import compare_direction
    dims, grids = CASES[case]()
    L = oracle.layout(dims)
    n = len(grids)
    kkt = pr.make_kkt_batch(L, grids, 1, mode="dynamics")
    dx0 = pr.make_dx0(L, 1)
    ric = Records(L, "ric").zeros(1, n)
    d_ref = Records(L, "dir").zeros(1, n)
    kk = kkt.copy()
    oracle.riccati_sweep_batch(L, grids, kk, ric, d_ref, dx0=dx0)
    d = Records(L, "dir").zeros(n)
    lib = _emu()
    lib.scan_emu_forward.restype = C.c_int
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    # the forward pass reads Fxx, Fvu, Fx only, which the backward pass leaves untouched
    lv = lib.scan_emu_forward(dims.nv, dims.nu, dims.ns_max, grid_array(grids), n, P(kkt[0]), P(ric[0]), P(dx0[0]), P(d))
    assert lv == int(np.ceil(np.log2(n - 1)))
    worst = compare_direction(L, grids, d, d_ref[0], TOL_SCAN, "forward scan")
    print("forward scan worst rel err %.2e" % worst)


def test_emulated_scan_flags_non_spd_quu(oracle):
    dims, grids = CASES["anymal_trot_short"]()
    L = oracle.layout(dims)
    kkt = pr.make_kkt_batch(L, grids, 1)[0]
    K = Records(L, "kkt")
    K.f(kkt[3], "Quu")[...] = -np.eye(dims.nu)
    lib = _emu()
    ps = np.zeros((len(grids), lib.scan_emu_ps_stride(dims.nv)))
    stat = C.c_uint(0)
    lib.scan_emu_backward(dims.nv, dims.nu, dims.ns_max, grid_array(grids), len(grids),
                          kkt.ctypes.data_as(C.POINTER(C.c_double)), ps.ctypes.data_as(C.POINTER(C.c_double)),
                          C.byref(stat))
    assert stat.value & 1


STO_CASES = {
    "anymal_jump_sto": lambda: pr.config_anymal_jump_sto()[:2],
    "anymal_jump_sto_short": lambda: pr.config_anymal_jump_sto(N=12, dt=0.05)[:2],
    "icub32_jump_sto": lambda: (lambda d, g, *_: (d, g))(*pr.config_icub_jump(nv=32)),
}


@pytest.mark.parametrize("case", sorted(STO_CASES))
@pytest.mark.parametrize("mode", ["dynamics", "factory"])
def test_sto_vector_pass_bodies_match_serial_oracle(oracle, case, mode):
    """Grids WITH switching-time optimisation: "matrix scan + serial vector pass" (robotoc_amd/csrc/riccati_scan_sto.hpp), the
    kernel bodies compiled for the host.  The matrix half (P_i of the scan; K_i, M_i) is taken from the serial recursion -- it
    does not see the STO terms --, the preparation and the vector pass then have to reproduce s, k, m, Psi, Phi, psi, phi, T,
    W, mt, mt_next, the five scalars and the STOPolicy of every transition of the oracle's serial recursion."""
    from helpers import compare_riccati
    dims, grids = STO_CASES[case]()
    if not any(g.sto for g in grids):
        pytest.skip("the configuration carries no STO flags")
    L = oracle.layout(dims)
    n, nx = len(grids), 2 * dims.nv
    kkt = pr.make_kkt_batch(L, grids, 1, mode=mode)[0]
    R = Records(L, "ric")
    ric_ref = R.zeros(n)
    assert oracle.riccati_backward(L, grids, kkt.copy(), ric_ref) == 0
    lib = _emu()
    stride = lib.scan_emu_ps_stride(dims.nv)
    # the scan's value records: P of every grid point (the s half of the records is not read on STO grids)
    ps = np.zeros((n, stride))
    for i in range(n):
        ps[i, :nx * nx] = np.asarray(R.f(ric_ref[i], "P")).T.reshape(-1)
    # the matrix half of the records; everything the vector pass owns starts out as NaN
    ric = R.zeros(n)
    for i in range(n):
        for f in ("P", "K", "M"):
            R.f(ric[i], f)[...] = R.f(ric_ref[i], f)
        for f in ("s", "k", "m", "Psi", "Phi", "psi_x", "phi_x", "psi_u", "phi_u", "T", "W", "mt", "mt_next", "dtsdx", "scal"):
            R.f(ric[i], f)[...] = np.nan
    R.f(ric[n - 1], "s")[...] = R.f(ric_ref[n - 1], "s")
    lib.scan_emu_backward_sto.restype = C.c_int
    lib.scan_emu_backward_sto.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    rc = lib.scan_emu_backward_sto(dims.nv, dims.nu, dims.ns_max, grid_array(grids), n, kkt.ctypes.data, ps.ctypes.data, ric.ctypes.data, 0.1)
    assert rc == 0
    worst = compare_riccati(L, grids, ric, ric_ref, TOL_SCAN, "sto vector pass", check_sto=True)
    # the STOPolicy of every transition
    for i, g in enumerate(grids[:-1]):
        for f, sl in (("dtsdx", slice(None)), ("scal", slice(5, 7))):
            a, b = np.asarray(R.f(ric[i], f))[sl], np.asarray(R.f(ric_ref[i], f))[sl]
            ok = np.isfinite(b) & (b != 0)
            if ok.any():
                # (the factory data is ill conditioned: tests/test_gpu_parity.py::test_anymal_jump_sto_ill_conditioned)
                assert np.allclose(a[ok], b[ok], rtol=1e-8 if mode == "dynamics" else 1e-6, atol=1e-10), (i, f)
    print("sto vector pass (%s, %s): worst rel err %.2e" % (case, mode, worst))
