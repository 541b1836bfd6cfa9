#!/usr/bin/env python3
"""ANYmal jump with switching-time optimisation solved twice by the Python shell (and once with RTOC_OPT_BACKWARD_REGISTER = 0): are the
KKT-error histories bit-identical run to run?  (tests/test_cpp_solver.py compares the Python and the C++ shell bit for bit.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import problems_jump as pj

hists = []
for reg in (1, 1, 0, 0):
    solver, x0, info = pj.anymal_jump_sto_solver(batch=1)
    solver.ctx.set_backward_register(reg)
    st = solver.solve(0.0, x0)
    h = np.array([e[0] for e in st.kkt_error])
    hists.append(h)
    print("register", reg, "iterations", st.iter, "mesh refinement at", st.mesh_refinement_iter, "first 4 errors", h[:4])
    solver.close()
n = min(len(h) for h in hists)
print("register=1 run 0 vs run 1 bit-identical over %d iterations:" % n, np.array_equal(hists[0][:n], hists[1][:n]),
      "first difference at", int(np.argmax(hists[0][:n] != hists[1][:n])) if not np.array_equal(hists[0][:n], hists[1][:n]) else None)
print("register=0 run 0 vs run 1 bit-identical:", np.array_equal(hists[2][:n], hists[3][:n]))
print("register 1 vs 0 max rel diff over the first 12:", np.abs(hists[0][:12] / hists[2][:12] - 1).max())
