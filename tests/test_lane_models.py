"""The lane-level numpy models of the two register-chained kernels (tools/rv_model.py: riccati_backward_rv_kernel; tools/cond_model.py:
condense_rv_kernel) against the CPU oracle, on CPU: every 64-lane operand / accumulator layout the kernels rely on -- "the C layout of a
symmetric matrix is its A fragment", "C as A gives the transpose, C as B the matrix", rider columns, lane shifts -- stated with explicit
arrays and multiplied through an emulation of v_mfma_f64_16x16x4_f64.  What these prove is the index algebra, not the hardware."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name, capsys, argv=()):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    old = sys.argv
    sys.argv = [name] + list(argv)
    try:
        mod.main()
    finally:
        sys.argv = old
    return capsys.readouterr().out


def test_backward_register_kernel_lane_model(oracle, capsys):
    out = _run("rv_model", capsys)
    errs = [float(x) for x in re.findall(r"'(?:P|s|K|k|M|m)': ([0-9.]+(?:e[-+][0-9]+)?)", out)]   # per stage: P, s, K, k (M, m) vs the oracle
    assert "rv lane model: ok" in out and len(errs) >= 20 and max(errs) < 1e-12, out[-400:]


def test_condensation_register_kernel_lane_model(oracle, capsys):
    out = _run("cond_model", capsys)
    m = re.search(r"worst relative error ([0-9.]+e[-+][0-9]+)", out)
    assert m and float(m.group(1)) < 1e-12, out[-400:]
    assert out.count("fields agree with the oracle") == 5
