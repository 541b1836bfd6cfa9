"""Regression pins for the parity tests: tests/golden/parity_pins.json from the parity summary of a green run.

  RTOC_PARITY_PINS=0 python -m pytest tests -q -m "not gpu"     # here   -> gpurun_out/parity_summary.json
  python tools/make_parity_pins.py                               # merge the CPU-side tests
  gpurun -- 'RTOC_PARITY_PINS=0 python -m pytest tests -q -m gpu' # on the box, summary merged back
  python tools/make_parity_pins.py                               # merge the GPU-side tests

Per test and comparison kind (instance / seed numbers stripped) the pin is 10x the worst observed error, rounded up to
two significant digits, never above the tolerance the test asserts; comparisons that came out exactly equal get no pin
(their tolerance stays the bound).  tests/helpers.py: record_parity asserts against min(tolerance, max(pin, 1e-13)):
a pin never binds below a few hundred ulps."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_summary.json")
DST = os.path.join(ROOT, "tests", "golden", "parity_pins.json")
FACTOR = 10.0


def round_up(x, digits=2):
    e = math.floor(math.log10(x)) - (digits - 1)
    return math.ceil(x / 10.0 ** e) * 10.0 ** e


def main():
    rows = json.load(open(SRC))
    out = {"factor": FACTOR, "pins": {}}
    if os.path.exists(DST):
        out = json.load(open(DST))
    n = 0
    for r in rows:
        pins = {}
        for key, (obs, tol) in r["items"].items():
            if obs > 0.0:
                pins[key] = float("%.2g" % min(tol, round_up(FACTOR * obs)))
                n += 1
        if pins:
            out["pins"][r["test"]] = pins
    out["pins"] = dict(sorted(out["pins"].items()))
    with open(DST, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("%d pins from %d tests of %s merged; %d tests pinned in total" % (n, len(rows), SRC, len(out["pins"])))


if __name__ == "__main__":
    main()
