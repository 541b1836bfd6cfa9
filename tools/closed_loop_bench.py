"""Times bench.closed_loop_trot (the whole constrained ANYmal-trot solver iteration on the device) on its own.
usage: python tools/closed_loop_bench.py [batch] [--batch-only]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    out = {"batch": bench.closed_loop_trot(0, batch)}
    if "--batch-only" not in sys.argv:
        out["single_instance"] = bench.closed_loop_trot(0, 1)
    print(json.dumps(out, indent=1))
