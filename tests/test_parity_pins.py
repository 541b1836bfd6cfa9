"""The regression pins (tests/golden/parity_pins.json) stay attached to tests that exist, and the recorder honours them."""
import ast
import json
import os
import re

import pytest

import helpers

HERE = os.path.dirname(os.path.abspath(__file__))


def _test_functions(path):
    with open(path) as f:
        tree = ast.parse(f.read())
    return {n.name for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name.startswith("test")}


def test_every_pin_belongs_to_a_test_that_exists():
    """A renamed or deleted test must not leave its pins behind (they would silently stop holding anything)."""
    with open(os.path.join(HERE, "golden", "parity_pins.json")) as f:
        doc = json.load(f)
    assert doc["factor"] == 10.0
    pins = doc["pins"]
    assert len(pins) > 150  # CPU-side and GPU-side tests both recorded
    funcs = {}
    for nodeid, kinds in pins.items():
        m = re.match(r"tests/(test_\w+\.py)::(\w+)(\[.*\])?$", nodeid)
        assert m, nodeid
        fn = funcs.setdefault(m.group(1), _test_functions(os.path.join(HERE, m.group(1))))
        assert m.group(2) in fn, "pin for a test that no longer exists: %s" % nodeid
        for kind, bound in kinds.items():
            assert kind == helpers.parity_key(kind), (nodeid, kind)  # keys are stored without instance / seed numbers
            assert 0.0 < bound <= 1e-3, (nodeid, kind, bound)


def test_recorder_asserts_the_pin_not_only_the_tolerance(monkeypatch):
    """record_parity holds a comparison to min(tolerance, pin): an error inside the written tolerance but 10x above what the
    case recorded fails."""
    node = "tests/test_x.py::test_y[case]"
    saved = (helpers._PINS[0], helpers.CURRENT_TEST[0])
    monkeypatch.setattr(helpers, "PARITY", {})
    try:
        helpers._PINS[0] = {node: {"riccati": 3e-12}}
        helpers.CURRENT_TEST[0] = node
        helpers.record_parity("riccati inst 3", 2.9e-12, 1e-9)          # inside the pin
        helpers.record_parity("direction inst 3", 5e-10, 1e-9)          # no pin for this kind: the tolerance holds
        with pytest.raises(AssertionError, match="regression pin"):
            helpers.record_parity("riccati inst 7", 4e-12, 1e-9)        # inside 1e-9, outside the pin
        assert helpers.PARITY[node][0][2] == 3e-12 and helpers.PARITY[node][1][2] == 1e-9
    finally:
        helpers._PINS[0], helpers.CURRENT_TEST[0] = saved
