"""Per-phase cycle breakdown (s_memtime stamps of lane 0) of one step of the serial vector pass of the scan on grids with
switching-time optimisation (riccati_scan_sto.hpp).  Needs the PROF build:
  make -C robotoc_amd/csrc PROF=1 OUT=../librtoc_hip_prof.so BUILD=build_prof ; RTOC_HIP_LIB=.../librtoc_hip_prof.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0
d, g, _ = pr.config_anymal_jump_sto()
c = capi.Context(d, len(g), 1, 0); c.set_grid(g)
c.upload(BUF_KKT, pr.make_kkt_batch(c.L, g, 1)); c.upload(BUF_DX0, pr.make_dx0(c.L, 1))
c.set_backward_scan(True)
capi.debug_profile(c)
for _ in range(3):
    c.riccati_backward(); c.sync()
p = capi.debug_profile(c)
names = ["fetch issue", "flush / transition", "P1 (+ barrier)", "P2", "P3", "P4 scalars", "drop"]
print("ms", c.time_phase(0, 10))
for st in (38, 30, 25, 20, 12, 5):
    row = p[st]
    dd = np.diff(row[:8])
    print("grid point %2d type %d dims %d sto %d: step %5d cycles (to next %5d) | " % (st, g[st].type, g[st].dims, g[st].sto, row[7] - row[0], p[st - 1][0] - row[0])
          + " | ".join("%s %d" % (n, x) for n, x in zip(names, dd))
          + " || last wave: flush done +%d, P1 rows done +%d; wave 0 rows done +%d, wave 1 +%d (from lane 0's P1 start)" % tuple(row[k] - row[2] for k in (10, 11, 12, 13)))
