"""Shared helpers for the parity tests."""
import numpy as np

from robotoc_amd.types import GRID_IMPACT, GRID_TERMINAL, Records

# ---- observed parity errors, for the record (tests/conftest.py prints them in the terminal summary and writes
#      gpurun_out/parity_summary.json): every comparison registers (label, worst observed error, asserted tolerance) under
#      the test that ran it, so that the log of a green run says how far inside its bounds it was ----
PARITY = {}
CURRENT_TEST = [None]


def record_parity(label, observed, tol):
    PARITY.setdefault(CURRENT_TEST[0] or "?", []).append((str(label), float(observed), float(tol)))


def check_parity(label, observed, tol):
    """Register an observed error and assert it against its tolerance."""
    record_parity(label, observed, tol)
    assert observed <= tol, "%s: observed %.3e > tolerance %.1e" % (label, observed, tol)
    return observed


def rel_err(a, b, floor=1e-300):
    """Relative Frobenius error ||a-b|| / max(||a||, ||b||, floor) (Eigen isApprox-style).
    `floor` guards quantities that are mathematically zero (e.g. W, mt_next when the switching
    constraint has as many rows as controls) -- the reference's own test skips those with
    `if (!T.isZero())` (test/riccati/riccati_factorizer_test.cpp:217-222)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    nb = np.linalg.norm(b)
    na = np.linalg.norm(a)
    den = max(nb, na, floor)
    return float(np.linalg.norm(a - b) / den)


def compare_riccati(L, grids, ric_gpu, ric_ref, tol, what="", check_sto=True):
    """Compare every meaningful field of the Riccati records stage by stage.
    Returns the worst relative error; raises AssertionError with a per-field report on failure."""
    R = Records(L, "ric")
    worst = 0.0
    bad = []
    N = len(grids) - 1
    for i, g in enumerate(grids):
        fields = ["P", "s"]
        if i < N and g.type != GRID_IMPACT:
            fields += ["K", "k"]
        if i < N and g.dims > 0:
            fields += ["M", "m"]
        if check_sto and i < N and g.sto:
            fields += ["Psi", "Phi"]
            if g.type != GRID_IMPACT:
                fields += ["T", "W", "psi_x", "psi_u"]
                if g.dims > 0:
                    fields += ["mt", "mt_next"]
        for f in fields:
            a = R.f(ric_gpu[i], f)
            b = R.f(ric_ref[i], f)
            if f in ("M",):
                a, b = a[:g.dims], b[:g.dims]
            if f in ("m", "mt", "mt_next"):
                a, b = a[:g.dims], b[:g.dims]
            floor = 1.0 if f in ("T", "W", "mt", "mt_next") else 1e-300
            e = rel_err(a, b, floor)
            worst = max(worst, e)
            if not (e <= tol):
                bad.append((i, f, e))
        if check_sto and i < N and g.sto:
            a = R.f(ric_gpu[i], "scal")[:5]
            b = R.f(ric_ref[i], "scal")[:5]
            scale = max(np.abs(b).max(), 1.0)
            e = float(np.abs(a - b).max() / scale)
            worst = max(worst, e)
            if not (e <= tol):
                bad.append((i, "scal", e))
    record_parity("riccati " + what, worst, tol)
    assert not bad, "%s riccati mismatch (stage, field, rel_err): %s" % (what, bad[:12])
    return worst


def compare_direction(L, grids, d_gpu, d_ref, tol, what=""):
    D = Records(L, "dir")
    worst = 0.0
    bad = []
    N = len(grids) - 1
    for i, g in enumerate(grids):
        fields = ["dx", "dlmdgmm"]
        if i < N and g.type != GRID_IMPACT:
            fields.append("du")
        for f in fields:
            e = rel_err(D.f(d_gpu[i], f), D.f(d_ref[i], f))
            worst = max(worst, e)
            if not (e <= tol):
                bad.append((i, f, e))
        if i < N and g.switching_constraint and g.dims > 0:
            e = rel_err(D.f(d_gpu[i], "dxi")[:g.dims], D.f(d_ref[i], "dxi")[:g.dims])
            worst = max(worst, e)
            if not (e <= tol):
                bad.append((i, "dxi", e))
        a = D.f(d_gpu[i], "dts")[:2]
        b = D.f(d_ref[i], "dts")[:2]
        e = float(np.abs(a - b).max() / max(np.abs(b).max(), 1.0))
        worst = max(worst, e)
        if not (e <= tol):
            bad.append((i, "dts", e))
    record_parity("direction " + what, worst, tol)
    assert not bad, "%s direction mismatch (stage, field, rel_err): %s" % (what, bad[:12])
    return worst


def _rel_err_rows(a, b, floor=1e-300):
    """rel_err per leading index: a, b are [batch, ...]; returns [batch]."""
    a = np.asarray(a, dtype=np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, dtype=np.float64).reshape(b.shape[0], -1)
    den = np.maximum(np.maximum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)), floor)
    return np.linalg.norm(a - b, axis=1) / den


def compare_batch(L, grids, ric_gpu, ric_ref, d_gpu, d_ref, tol, what="", check_sto=True):
    """compare_riccati + compare_direction for whole batches ([batch, stages, stride] arrays), vectorised over
    the instance axis: the same per-instance, per-stage, per-field relative errors (so that thousands of
    instances are checked in seconds).  Returns the worst error; raises with (instance, stage, field) on failure."""
    R, D = Records(L, "ric"), Records(L, "dir")
    worst, bad = 0.0, []
    N = len(grids) - 1

    def chk(i, f, e):
        nonlocal worst
        w = float(e.max())
        worst = max(worst, w)
        if not (w <= tol):
            bad.append((int(e.argmax()), i, f, w))
    for i, g in enumerate(grids):
        fields = ["P", "s"]
        if i < N and g.type != GRID_IMPACT:
            fields += ["K", "k"]
        if i < N and g.dims > 0:
            fields += ["M", "m"]
        if check_sto and i < N and g.sto:
            fields += ["Psi", "Phi"]
            if g.type != GRID_IMPACT:
                fields += ["T", "W", "psi_x", "psi_u"]
                if g.dims > 0:
                    fields += ["mt", "mt_next"]
        for f in fields:
            a, b = R.f(ric_gpu[:, i], f), R.f(ric_ref[:, i], f)
            if f in ("M", "m", "mt", "mt_next"):
                a, b = a[:, :g.dims], b[:, :g.dims]
            chk(i, f, _rel_err_rows(a, b, 1.0 if f in ("T", "W", "mt", "mt_next") else 1e-300))
        if check_sto and i < N and g.sto:
            a, b = R.f(ric_gpu[:, i], "scal")[:, :5], R.f(ric_ref[:, i], "scal")[:, :5]
            chk(i, "scal", np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1.0))
        if d_gpu is None:
            continue
        fields = ["dx", "dlmdgmm"]
        if i < N and g.type != GRID_IMPACT:
            fields.append("du")
        for f in fields:
            chk(i, f, _rel_err_rows(D.f(d_gpu[:, i], f), D.f(d_ref[:, i], f)))
        if i < N and g.switching_constraint and g.dims > 0:
            chk(i, "dxi", _rel_err_rows(D.f(d_gpu[:, i], "dxi")[:, :g.dims], D.f(d_ref[:, i], "dxi")[:, :g.dims]))
        a, b = D.f(d_gpu[:, i], "dts")[:, :2], D.f(d_ref[:, i], "dts")[:, :2]
        chk(i, "dts", np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1.0))
    record_parity("batch " + what, worst, tol)
    assert not bad, "%s mismatch (instance, stage, field, rel_err): %s" % (what, bad[:12])
    return worst
