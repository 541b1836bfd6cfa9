import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """On a GPU box: bring up torch's HIP runtime BEFORE the first rtoc context of the process.  torch ships its own copy of the
    HIP runtime; the tests that hold device tensors (tests/test_determinism.py, the RCCL gather) initialise it lazily, and after
    ~70 tests' worth of contexts of the library's runtime that late initialisation was seen to fail with "No HIP GPUs are
    available" (order-dependent: the same tests pass on their own and in the full suite's order).  bench.py has torch first too."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(autouse=True)
def _parity_scope(request):
    import helpers
    helpers.CURRENT_TEST[0] = request.node.nodeid
    yield
    helpers.CURRENT_TEST[0] = None


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Observed parity errors of the run (tests/helpers.py: record_parity): per test the comparison that came closest to its
    tolerance.  Printed under -q as well, so that the driver's log of a green run carries numbers, and written to
    gpurun_out/parity_summary.json (merged back from the GPU box)."""
    import json
    import re
    import helpers
    if not helpers.PARITY:
        return
    rows = []
    for test, items in helpers.PARITY.items():
        label, obs, tol = max(items, key=lambda it: it[1] / it[2] if it[2] > 0 else 0.0)
        agg = {}
        for lb, ob, tl in items:   # per comparison kind (instance / seed numbers stripped): worst observed, its bound
            key = helpers.parity_key(lb)
            if key not in agg or ob > agg[key][0]:
                agg[key] = [ob, tl]
        rows.append({"test": test, "checks": len(items), "closest": label.strip(), "observed": obs, "tolerance": tol,
                     "worst_observed": max(it[1] for it in items), "loosest_tolerance": max(it[2] for it in items), "items": agg})
    rows.sort(key=lambda r: -(r["observed"] / r["tolerance"] if r["tolerance"] > 0 else 0.0))
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_summary.json"), "w") as f:
            json.dump(rows, f, indent=1)
    except OSError:
        pass
    tr = terminalreporter
    tr.write_line("")
    pinned = sum(len(v) for v in helpers.parity_pins().values())
    tr.write_line("PARITY: %d tests registered %d comparisons (observed error / bound = min(asserted tolerance, regression pin); "
                  "closest to its bound first; %d pins loaded)" % (len(rows), sum(r["checks"] for r in rows), pinned))
    for r in rows[:45]:
        name = r["test"].split("tests/")[-1]
        tr.write_line("  %-98s %9.2e / %7.0e  [%s]" % (name[:98], r["observed"], r["tolerance"], r["closest"][:40]))
    if len(rows) > 45:
        rest = rows[45:]
        tr.write_line("  ... %d more tests, every one below %.1e of its tolerance" % (len(rest), max(r["observed"] / r["tolerance"] for r in rest)))
    loose = [r for r in rows if r["observed"] > 0 and r["tolerance"] / r["observed"] > 1e3]
    tr.write_line("PARITY: largest observed error %.2e; %d tests run more than 1000x inside their tolerance"
                  % (max(r["worst_observed"] for r in rows), len(loose)))
