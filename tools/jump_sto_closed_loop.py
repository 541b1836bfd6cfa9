"""BASELINE configs[2] closed on the device: the ANYmal jump of examples/anymal/python/jump_sto.py (stand, flight, stand; both
events with switching-time optimisation; ConfigurationSpaceCost; the example's Constraints object; minimum dwell times) through
robotoc_amd.solver.OCPSolver -- KKT error and event times per iteration.  Usage: jump_sto_closed_loop.py [batch] [variant]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from robotoc_amd import problems_jump as pj


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    variant = sys.argv[2] if len(sys.argv) > 2 else "example"
    kw = {}
    if variant == "free":
        kw = dict(with_limits=False, with_cones=False)
    elif variant == "nolimits":
        kw = dict(with_limits=False)
    kw["horizon_scan"] = os.environ.get("RTOC_HORIZON_SCAN", "off")
    if variant.startswith("icub"):   # BASELINE configs[3] as examples/icub/python/jump_sto.py poses it; icub1: its first jump only
        kw = dict(horizon_scan=kw["horizon_scan"], jumps=1 if variant == "icub1" else 2)
        if len(sys.argv) > 3:
            kw["max_iter"] = int(sys.argv[3])
        solver, x0, info = pj.icub_jump_sto_solver(batch=batch, **kw)
    else:
        solver, x0, info = pj.anymal_jump_sto_solver(batch=batch, **kw)
    t0 = time.perf_counter()
    st = solver.solve(0.0, x0)
    dt = time.perf_counter() - t0
    for it, e in enumerate(st.kkt_error):
        ts = st.ts[it][0] if st.ts else []
        print("iter %3d  KKT %.3e (worst of %d)  ts[0] %s%s" % (it, float(np.max(e)), batch, np.array2string(np.asarray(ts), precision=4),
                                                                "  <- mesh refinement" if it + 1 in st.mesh_refinement_iter else ""))
    print("converged", st.convergence, "iterations", st.iter, "event times", solver.event_times[0], "mesh refinements at", st.mesh_refinement_iter,
          "%.1f ms per iteration" % (1e3 * dt / max(st.iter, 1)))
    print("status", np.unique(solver.ctx.status()))
    solver.close()


if __name__ == "__main__":
    main()
