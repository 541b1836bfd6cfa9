"""CPU-side checks: the C-ABI library loads and exports every symbol include/rtoc.h and include/rtoc_robot.h declare
(no compute calls without a GPU), layouts agree between product and oracle, grid construction."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from robotoc_amd import capi, grid as G, problems as pr
from robotoc_amd.types import (GRID_IMPACT, GRID_INTERMEDIATE, GRID_LIFT, GRID_TERMINAL, Dims,
                               anymal_dims, icub_dims, iiwa14_dims)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    capi.build()
    lib = C.CDLL(capi.lib_path())
    hdr = open(os.path.join(ROOT, "include", "rtoc.h")).read() + open(os.path.join(ROOT, "include", "rtoc_robot.h")).read()
    declared = set(re.findall(r"\b(rtoc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"rtoc_compute_layout", "rtoc_cone_stride", "rtoc_cone_dgdf_off", "rtoc_wrench_cone_stride"}  # static inline in rtoc_layout.h
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export %s" % name
    for name in capi.EXPORTS:
        assert name in declared


def test_no_gpu_means_loud_failure_not_fallback():
    lib = capi.lib()
    if lib.rtoc_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.RtocError):
        capi.Context(anymal_dims(), 8, 1, 0)


def test_layout_product_equals_oracle(oracle):
    for dims in (anymal_dims(), icub_dims(35), icub_dims(32), iiwa14_dims()):
        a, b = capi.layout_for(dims), oracle.layout(dims)
        assert bytes(a) == bytes(b)
        for rl in (a.kkt, a.ric, a.dir, a.cdd):
            offs = list(rl.off)[:rl.nfields]
            assert all(o % 8 == 0 for o in offs) and rl.stride % 8 == 0  # 64 B aligned fields
            assert offs == sorted(offs)


def test_dims_supported_table():
    lib = capi.lib()
    for dims in (anymal_dims(), icub_dims(35), icub_dims(32), iiwa14_dims()):
        assert lib.rtoc_dims_supported(C.byref(dims)) == 1
    assert lib.rtoc_dims_supported(C.byref(Dims(5, 5, 0, 0, 0, 0))) == 0


def test_trot_grid_structure():
    """47 grids = 41 + 2 lifts + 2*2 impact grids (time_discretization.cpp:45); switching constraint
    two grids before each impact (:139-141); dt of split intervals adds up."""
    dims, grids, _ = pr.config_anymal_trot()
    types = [g.type for g in grids]
    assert len(grids) == 47 and types[-1] == GRID_TERMINAL
    assert types.count(GRID_LIFT) == 2 and types.count(GRID_IMPACT) == 2
    for i, g in enumerate(grids):
        if g.type == GRID_IMPACT:
            assert g.dt == 0.0 and grids[i - 2].switching_constraint == 1 and grids[i - 2].dims == 6
            assert g.time_stage == -1
        if g.switching_constraint:
            assert grids[i + 2].type == GRID_IMPACT
    assert abs(sum(g.dt for g in grids) - 0.8) < 1e-12
    assert not any(g.sto for g in grids)


def test_jump_sto_grid_flags():
    dims, grids, _ = pr.config_anymal_jump_sto()
    assert len(grids) == 44
    lift = [i for i, g in enumerate(grids) if g.type == GRID_LIFT][0]
    imp = [i for i, g in enumerate(grids) if g.type == GRID_IMPACT][0]
    assert all(g.sto for g in grids[:imp + 1]) and grids[imp - 2].dims == 12
    assert all(grids[i].sto_next for i in range(lift, imp))
    assert not grids[-1].sto
    # phase-based time steps: uniform inside a phase (correctTimeSteps)
    d0 = [g.dt for g in grids[:lift]]
    assert max(d0) - min(d0) < 1e-12


def test_uniform_grid():
    g = G.uniform_grid(20, 0.05)
    assert len(g) == 21 and all(x.type == GRID_INTERMEDIATE for x in g[:-1])


def test_bench_counter_pass_is_refused_unless_measured_on_or_carried_over_to_these_sources(monkeypatch):
    """bench.py: a committed counter pass (traffic, fp64 flops) counts only if it carries the hash of the library's kernel
    sources, or names that hash in a hand-written carry-over entry with its justification -- which then appears in the line."""
    import bench
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "aaaa")
    ok, note = bench.counter_pass_valid({"_kernel_source_hash": "aaaa"})
    assert ok and "matches" in note
    ok, note = bench.counter_pass_valid({"_kernel_source_hash": "bbbb"})
    assert not ok and "stale" in note
    ok, note = bench.counter_pass_valid({"_kernel_source_hash": "bbbb", "_carried_over": [{"to": "cccc", "change": "x"}]})
    assert not ok
    ok, note = bench.counter_pass_valid({"_kernel_source_hash": "bbbb", "_carried_over": [{"to": "aaaa", "change": "flag wait moved"}]})
    assert ok and "CARRIED OVER" in note and "flag wait moved" in note and "bbbb" in note
