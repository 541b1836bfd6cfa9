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
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
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
