"""Kernel specialisation coverage: every robot shape of the SHAPES list (robotoc_amd/csrc/Makefile) is reported by
rtoc_dims_supported, unlisted ones are refused with RTOC_ERR_UNSUPPORTED_DIMS; and a shape that is neither a
quadruped / humanoid nor the contact-free arm -- the reference's test manipulator, iiwa14 with ONE point contact
(test/test_helper/robot_factory.cpp:12-24: nv = nu = 7, max_dimf = 3, fixed base) -- runs the whole hot path against the
oracle: contact -> lift -> flight -> impact (with its switching-constraint grid) -> contact."""
import ctypes as C

import numpy as np
import pytest

from helpers import compare_direction, compare_riccati, rel_err
from robotoc_amd import problems as pr
from robotoc_amd.grid import ContactSequence, Event, discretize
from robotoc_amd.types import BUF_CDD, BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, Dims, GRID_IMPACT, GRID_LIFT, Records


def test_every_listed_shape_is_supported_and_others_are_not():
    from robotoc_amd import capi
    shapes = capi.compiled_shapes()
    assert (18, 12, 12) in shapes and (7, 7, 3) in shapes and len(shapes) >= 5
    lib = capi.lib()
    for nv, nu, ns in shapes:
        assert lib.rtoc_dims_supported(C.byref(Dims(nv, nu, nv - nu, ns, ns, 0))) == 1, (nv, nu, ns)
    # no kernel set and no plugin (and no RTOC_SHAPE_JIT in the environment): refused; nf_max != ns_max / np != nv - nu: always
    for bad in (Dims(11, 11, 0, 0, 0, 0), Dims(18, 12, 6, 12, 6, 0), Dims(18, 12, 0, 12, 12, 0)):
        assert lib.rtoc_dims_supported(C.byref(bad)) == 0
    # layout of any dims is available without a kernel set (a host can size its buffers first)
    L = capi.layout_for(Dims(12, 12, 0, 0, 0, 0))
    assert L.nx == 24 and L.kkt.stride > 0


def manipulator_with_contact(N=14, dt=0.02):
    dims = Dims(7, 7, 0, 3, 3, 48)
    cs = ContactSequence([3, 0, 3], [Event("lift", 0.07, sto=False), Event("impact", 0.15, sto=False, impact_dimf=3)])
    return dims, discretize(N, N * dt, 0.0, cs)


def test_manipulator_grid_has_every_kind():
    _, grids = manipulator_with_contact()
    assert any(g.type == GRID_IMPACT for g in grids) and any(g.type == GRID_LIFT for g in grids)
    assert any(g.dims == 3 for g in grids) and {g.dimf for g in grids} == {0, 3}


@pytest.mark.gpu
@pytest.mark.parametrize("scan", [False, True])
def test_manipulator_with_one_point_contact_sweep(oracle, scan):
    from robotoc_amd import capi
    dims, grids = manipulator_with_contact()
    batch = 5
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        ctx.set_backward_scan(scan)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="factory")
        dx0 = pr.make_dx0(L, batch)
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_DX0, dx0)
        ctx.riccati_sweep()
        assert (ctx.status() == 0).all()
        ric, d = ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir")
        ric_ref, d_ref = Records(L, "ric").zeros(batch, len(grids)), Records(L, "dir").zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        tol = 1e-8 if scan else 1e-9
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric_ref[b], tol, "manipulator inst %d" % b)
            compare_direction(L, grids, d[b], d_ref[b], tol, "manipulator inst %d" % b)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_manipulator_with_one_point_contact_sqp_hot_path(oracle):
    from robotoc_amd import capi
    dims, grids = manipulator_with_contact()
    batch = 3
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt, cdd = pr.make_precondense_batch(L, grids, batch)
        dx0 = pr.make_dx0(L, batch)
        for buf, arr in ((BUF_KKT, kkt), (BUF_CDD, cdd), (BUF_DX0, dx0)):
            ctx.upload(buf, arr)
        ctx.condense()
        kkt_gpu = ctx.download_records(BUF_KKT, "kkt")
        ctx.riccati_sweep()
        ctx.expand(0.995)
        assert (ctx.status() == 0).all()
        d_gpu = ctx.download_records(BUF_DIR, "dir")
        kk, cc = kkt.copy(), cdd.copy()
        assert (oracle.condense_batch(L, grids, kk, cc) == 0).all()
        K, D = Records(L, "kkt"), Records(L, "dir")
        for f in ("Qxx", "Qxu", "Quu", "lx", "lu", "Fxx", "Fvu", "Fx", "Phix", "Phiu", "Pres"):
            assert rel_err(K.f(kkt_gpu, f), K.f(kk, f)) < 1e-9, f
        ric_ref, d_ref = Records(L, "ric").zeros(batch, len(grids)), D.zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kk, ric_ref, d_ref, dx0=dx0)
        oracle.expand_batch(L, grids, cc, d_ref)
        for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu"):
            from helpers import check_parity
            check_parity("sqp directions " + f, rel_err(D.f(d_gpu, f), D.f(d_ref, f)), 1e-9)
    finally:
        ctx.close()


# ---- shapes beyond the compiled-in list: plugins (make plugin / capi.build_plugin / RTOC_SHAPE_JIT=1) ----
PLUGIN_SHAPES = [(12, 6, 6), (24, 18, 12)]  # a small quadruped-like floating base; a biped-like one with a 48-wide state (no role-split kernel: nu > 13)


def test_plugin_shapes_build_and_are_picked_up():
    from robotoc_amd import capi
    lib = capi.lib()
    for nv, nu, ns in PLUGIN_SHAPES:
        capi.build_plugin(nv, nu, ns)
        assert lib.rtoc_dims_supported(C.byref(Dims(nv, nu, nv - nu, ns, ns, 0))) == 1, (nv, nu, ns)


def test_library_builds_a_missing_shape_itself_when_asked(tmp_path):
    """RTOC_SHAPE_JIT=1: rtoc_dims_supported / rtoc_create run `make plugin` for a shape nobody built (needs hipcc and the
    source directory the library was built from; a fresh process, because the switch is read from the environment)."""
    import os
    import subprocess
    import sys
    from robotoc_amd import capi
    so = os.path.join(os.path.dirname(capi.lib_path()), "librtoc_shape_9_9_0.so")
    if os.path.exists(so):
        os.remove(so)
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from robotoc_amd import capi; from robotoc_amd.types import Dims; "
            "print(capi.lib().rtoc_dims_supported(C.byref(Dims(9, 9, 0, 0, 0, 0))))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    off = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "RTOC_SHAPE_JIT": "0"}, timeout=300)
    assert off.stdout.strip().endswith("0"), (off.stdout, off.stderr)
    on = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "RTOC_SHAPE_JIT": "1"}, timeout=300)
    assert on.stdout.strip().endswith("1"), (on.stdout, on.stderr)
    assert os.path.exists(so)
    os.remove(so)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", PLUGIN_SHAPES)
def test_plugin_shape_runs_the_hot_path_against_the_oracle(oracle, shape):
    """lift / impact / switching-constraint grids on a robot shape that is NOT in the library: sweep and condensation + expansion"""
    from robotoc_amd import capi
    nv, nu, ns = shape
    capi.build_plugin(nv, nu, ns)
    dims = Dims(nv, nu, nv - nu, ns, ns, 0)
    half = ns // 2
    cs = ContactSequence([ns, half, ns], [Event("lift", 0.07, sto=False), Event("impact", 0.15, sto=False, impact_dimf=ns - half)])
    grids = discretize(14, 14 * 0.02, 0.0, cs)
    batch = 4
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L = ctx.L
        ctx.set_grid(grids)
        kkt = pr.make_kkt_batch(L, grids, batch, mode="factory")
        dx0 = pr.make_dx0(L, batch)
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_DX0, dx0)
        ctx.riccati_sweep()
        assert (ctx.status() == 0).all()
        ric, d = ctx.download_records(BUF_RIC, "ric"), ctx.download_records(BUF_DIR, "dir")
        ric_ref, d_ref = Records(L, "ric").zeros(batch, len(grids)), Records(L, "dir").zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kkt.copy(), ric_ref, d_ref, dx0=dx0)
        for b in range(batch):
            compare_riccati(L, grids, ric[b], ric_ref[b], 1e-9, "plugin %s inst %d" % (shape, b))
            compare_direction(L, grids, d[b], d_ref[b], 1e-9, "plugin %s inst %d" % (shape, b))
        # the whole hot path downstream of the linearisation
        kkt, cdd = pr.make_precondense_batch(L, grids, batch)
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_CDD, cdd)
        ctx.upload(BUF_DX0, dx0)
        ctx.condense()
        got_kkt = ctx.download_records(BUF_KKT, "kkt")
        ctx.riccati_sweep()
        ctx.expand(0.995)
        assert (ctx.status() == 0).all()
        d_gpu = ctx.download_records(BUF_DIR, "dir")
        kkt_ref, cdd_ref = kkt.copy(), cdd.copy()
        assert (oracle.condense_batch(L, grids, kkt_ref, cdd_ref) == 0).all()
        K, D = Records(L, "kkt"), Records(L, "dir")
        for f in ("Qxx", "Qxu", "Quu", "lx", "lu", "Fxx", "Fvu", "Fx", "Phix", "Phiu", "Pres"):
            assert rel_err(K.f(got_kkt, f), K.f(kkt_ref, f)) < 1e-9, f
        ric_ref, d_ref = Records(L, "ric").zeros(batch, len(grids)), D.zeros(batch, len(grids))
        oracle.riccati_sweep_batch(L, grids, kkt_ref, ric_ref, d_ref, dx0=dx0)
        oracle.expand_batch(L, grids, cdd_ref, d_ref)
        for f in ("dx", "du", "dlmdgmm", "daf", "dbetamu"):
            from helpers import check_parity
            check_parity("sqp directions " + f, rel_err(D.f(d_gpu, f), D.f(d_ref, f)), 1e-9)
    finally:
        ctx.close()
