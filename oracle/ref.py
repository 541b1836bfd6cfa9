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


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def riccati_sweep(L, grids, kkt, ric, dirs, max_dts0=0.1, contact_dim=3, forward=True):
    """RiccatiRecursion::backward (+ forward) RiccatiRecursion of ONE instance ([stages, stride] arrays); kkt is
    mutated in place like the reference's containers, dirs[0].dx is the input."""
    lib().ref_riccati_sweep(C.byref(L), grid_array(grids), len(grids), _p(kkt), _p(ric), _p(dirs), max_dts0,
                            contact_dim, int(forward))


def unconstr_sweep(L, nstages, dt, kkt, ric, dirs, forward=True):
    lib().ref_unconstr_sweep(C.byref(L), nstages, dt, _p(kkt), _p(ric), _p(dirs), int(forward))


def condense_stage(L, g, kkt_rec, cdd_rec, damping=0.0, contact_dim=3):
    """condenseContactDynamics / condenseImpactDynamics of one grid point (no evalKKT-tail scalings)."""
    lib().ref_condense_stage(C.byref(L), C.byref(g), _p(kkt_rec), _p(cdd_rec), damping, contact_dim)


def expand_stage(L, g, cdd_rec, dir_rec, dir_next_rec, contact_dim=3):
    lib().ref_expand_stage(C.byref(L), C.byref(g), _p(cdd_rec), _p(dir_rec), _p(dir_next_rec), contact_dim)


def correct_costate(L, se3_rec, dir_rec):
    lib().ref_correct_costate(C.byref(L), _p(se3_rec), _p(dir_rec))
