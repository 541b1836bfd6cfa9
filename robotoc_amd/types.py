"""ctypes mirrors of include/rtoc_layout.h and numpy views on packed records.

The record layout itself is computed in C (rtoc_compute_layout); Python only
mirrors the structs so that tests / bench can build and inspect packed buffers.
Field semantics follow the reference containers cited in rtoc_layout.h.
"""
import ctypes as C

import numpy as np

GRID_INTERMEDIATE, GRID_IMPACT, GRID_LIFT, GRID_TERMINAL = 0, 1, 2, 3

KKT_FIELDS = ["Fxx", "Fvu", "Qxx", "Qxu", "Quu", "Fx", "lx", "lu", "fx", "hx", "hu", "scal",
              "Phix", "Phiu", "Phit", "Pres"]
RIC_FIELDS = ["P", "s", "Psi", "Phi", "psi_x", "phi_x", "psi_u", "phi_u", "scal", "K", "k", "T",
              "W", "M", "m", "mt", "mt_next", "dtsdx"]
DIR_FIELDS = ["dx", "du", "dlmdgmm", "dxi", "dts", "daf", "dbetamu", "dnu_passive"]
CDD_FIELDS = ["dIDda", "dIDCdqv", "dCda", "IDC", "Qaa", "Qff", "Qqf", "la", "lf", "ha", "hf",
              "Phia", "lu_passive", "MJtJinv", "MJtJinv_dIDCdqv", "MJtJinv_IDC", "Qafqv",
              "Qafu_full", "laf", "Qxu_passive", "Quu_passive_topRight", "haf"]

STAT_QUU_NOT_SPD, STAT_S_NOT_SPD, STAT_NAN, STAT_M_NOT_SPD, STAT_FXX_UNSTRUCTURED = 1, 2, 4, 8, 16

BUF_KKT, BUF_RIC, BUF_DIR, BUF_CDD, BUF_CON, BUF_DX0, BUF_STEP, BUF_SE3, BUF_CONE, BUF_SOL = range(10)
SOL_FIELDS = ["q", "v", "a", "u", "f", "lmd", "gmm", "beta", "mu", "nu_passive", "xi"]
SE3_STRIDE, SE3_FQQ_INV, SE3_FQQ_PREV_INV = 72, 0, 36
OPT_WRITEBACK_KKT, OPT_MAX_DTS0, OPT_BACKWARD_WAVES, OPT_CONTACT_INV_DAMPING, OPT_SWEEP_CHUNKS, OPT_CONDENSE_SPLIT, OPT_BACKWARD_SCAN, OPT_CONDENSE_KEEP_QAF, OPT_FXX_STRUCTURE, OPT_GRAPH, OPT_SWITCHING_TRANSPORT, OPT_IMPACT_CONES, OPT_UNCONSTR_DENSE, OPT_LINEARIZE_FUSED, OPT_CONE_JACOBIAN, OPT_LINEARIZE_DOFS_PER_PASS, OPT_BACKWARD_REGISTER, OPT_CONDENSE_REGISTER = range(18)


class Dims(C.Structure):
    _fields_ = [("nv", C.c_int), ("nu", C.c_int), ("np", C.c_int), ("nf_max", C.c_int),
                ("ns_max", C.c_int), ("nc_max", C.c_int)]

    @property
    def nx(self):
        return 2 * self.nv


class Grid(C.Structure):
    _fields_ = [("type", C.c_int), ("sto", C.c_int), ("sto_next", C.c_int),
                ("switching_constraint", C.c_int), ("dimf", C.c_int), ("dims", C.c_int),
                ("num_grids_in_phase", C.c_int), ("time_stage", C.c_int), ("dt", C.c_double)]


class BoxRow(C.Structure):
    _fields_ = [("var", C.c_int), ("index", C.c_int), ("sign", C.c_int), ("level", C.c_int)]


VAR_Q, VAR_V, VAR_U, VAR_A = 0, 1, 2, 3
CON_FIELDS = ["slack", "dual", "residual", "cmpl", "cond", "dslack", "ddual"]


def joint_limit_rows(dims, acceleration=False):
    """The six joint-limit components of examples/anymal/trot.cpp:134-146, in the order they are
    added: position lower/upper (position level), velocity lower/upper (velocity level), torques
    lower/upper (acceleration level); each acts on the tail nu entries of q / v or on u.
    acceleration=True: JointAccelerationLowerLimit / UpperLimit after them (acceleration level, the tail nu entries of a;
    src/constraints/joint_acceleration_lower_limit.cpp) -- 8 nu rows."""
    rows = []
    npv, nu = dims.np, dims.nu
    for var, level in ((VAR_Q, 2), (VAR_V, 1), (VAR_U, 0)) + (((VAR_A, 0),) if acceleration else ()):
        for sign in (-1, +1):
            for j in range(nu):
                rows.append(BoxRow(var, j if var == VAR_U else npv + j, sign, level))
    return rows


class RecordLayout(C.Structure):
    _fields_ = [("off", C.c_int * 24), ("stride", C.c_int), ("nfields", C.c_int)]


class Layout(C.Structure):
    _fields_ = [("dims", Dims), ("nx", C.c_int), ("nvf_max", C.c_int), ("kkt", RecordLayout),
                ("ric", RecordLayout), ("dir", RecordLayout), ("cdd", RecordLayout),
                ("con", RecordLayout), ("sol", RecordLayout)]


def anymal_dims(nc_max=96):
    """ANYmal: nv=18, 12 actuated joints, 4 point contacts; nc_max = 72 joint-limit rows
    + 20 friction-cone rows, padded."""
    return Dims(18, 12, 6, 12, 12, nc_max)


def icub_dims(nv=35, nc_max=None):
    """iCub: nv=35 per the reference URDF (BASELINE.json names nv=32); 2 surface contacts."""
    return Dims(nv, nv - 6, 6, 12, 12, 6 * (nv - 6) if nc_max is None else nc_max)


def iiwa14_dims():
    return Dims(7, 7, 0, 0, 0, 0)


def grid_array(grids):
    arr = (Grid * len(grids))()
    for i, g in enumerate(grids):
        arr[i] = g
    return arr


def _shapes(L, which):
    d = L.dims
    nv, nu, nx, nf, ns, nvf = d.nv, d.nu, 2 * d.nv, d.nf_max, d.ns_max, d.nv + d.nf_max
    if which == "kkt":
        return dict(Fxx=(nx, nx), Fvu=(nv, nu), Qxx=(nx, nx), Qxu=(nx, nu), Quu=(nu, nu),
                    Fx=(nx,), lx=(nx,), lu=(nu,), fx=(nx,), hx=(nx,), hu=(nu,), scal=(8,),
                    Phix=(ns, nx), Phiu=(ns, nu), Phit=(ns,), Pres=(ns,))
    if which == "ric":
        return dict(P=(nx, nx), s=(nx,), Psi=(nx,), Phi=(nx,), psi_x=(nx,), phi_x=(nx,),
                    psi_u=(nu,), phi_u=(nu,), scal=(8,), K=(nx, nu), k=(nu,), T=(nu,), W=(nu,),
                    M=(ns, nx), m=(ns,), mt=(ns,), mt_next=(ns,), dtsdx=(nx,))
    if which == "dir":
        return dict(dx=(nx,), du=(nu,), dlmdgmm=(nx,), dxi=(ns,), dts=(8,), daf=(nvf,),
                    dbetamu=(nvf,), dnu_passive=(8,))
    if which == "cdd":
        return dict(dIDda=(nv, nv), dIDCdqv=(nvf, nx), dCda=(nf, nv), IDC=(nvf,), Qaa=(nv,),
                    Qff=(nf, nf), Qqf=(nv, nf), la=(nv,), lf=(nf,), ha=(nv,), hf=(nf,),
                    Phia=(ns, nv), lu_passive=(8,), MJtJinv=(nvf, nvf),
                    MJtJinv_dIDCdqv=(nvf, nx), MJtJinv_IDC=(nvf,), Qafqv=(nvf, nx),
                    Qafu_full=(nvf, nv), laf=(nvf,), Qxu_passive=(nx, 8),
                    Quu_passive_topRight=(8, nu), haf=(nvf,))
    if which == "con":
        return {f: (d.nc_max,) for f in CON_FIELDS}
    if which == "sol":
        return dict(q=(nv + 1,), v=(nv,), a=(nv,), u=(nu,), f=(nf,), lmd=(nv,), gmm=(nv,), beta=(nv,), mu=(nf,),
                    nu_passive=(8,), xi=(ns,))
    raise KeyError(which)


_NAMES = dict(kkt=KKT_FIELDS, ric=RIC_FIELDS, dir=DIR_FIELDS, cdd=CDD_FIELDS, con=CON_FIELDS, sol=SOL_FIELDS)


class Records:
    """numpy view helper over a packed buffer of shape [..., stride].

    ``rec.f(buf, "Qxx")`` returns a writable view of shape ``buf.shape[:-1] + (rows, cols)``
    in *column-major* element order exposed as a normal numpy array A[..., i, j]
    (i row, j column), i.e. strides are arranged so that A[..., i, j] addresses
    element i + j*rows of the packed field.  Note ``K`` is exposed as K^T
    (nx x nu column-major == the reference's row-major nu x nx K).
    """

    def __init__(self, L, which):
        self.L = L
        self.which = which
        self.rl = getattr(L, which)
        self.names = _NAMES[which]
        self.shapes = _shapes(L, which)
        self.stride = self.rl.stride

    def offset(self, name):
        return self.rl.off[self.names.index(name)]

    def f(self, buf, name):
        shp = self.shapes[name]
        o = self.offset(name)
        n = int(np.prod(shp))
        flat = buf[..., o:o + n]
        if len(shp) == 1:
            return flat
        r, c = shp
        v = flat.reshape(tuple(buf.shape[:-1]) + (c, r))  # a view (the last axis is split) -- numpy and torch alike
        return v.swapaxes(-1, -2)  # A[..., i, j] = flat[i + j*r]

    def zeros(self, *lead):
        return np.zeros(tuple(lead) + (self.stride,), dtype=np.float64)


def cone_dgdf_off(nv, max_contacts):
    """include/rtoc_layout.h: rtoc_cone_dgdf_off"""
    return (max_contacts * 5 * nv + 7) & ~7


def cone_stride(nv, max_contacts):
    """include/rtoc_layout.h: rtoc_cone_stride"""
    return cone_dgdf_off(nv, max_contacts) + ((max_contacts * 15 + 7) & ~7)


WRENCH_ROWS, FRICTION_ROWS = 17, 5


def wrench_cone_stride(max_contacts):
    """include/rtoc_layout.h: rtoc_wrench_cone_stride"""
    return (max_contacts * WRENCH_ROWS * 6 + 7) & ~7
