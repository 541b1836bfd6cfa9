"""Raw s_memtime stamps of the role-split backward kernel (PROF build): matrix wave slots 0..13, vector
wave slots 14..29, relative to the matrix wave's stage start.  Usage: phase_profile_rs.py [batch] [waves]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dims, grids, _ = pr.config_anymal_trot()
ctx = capi.Context(dims, len(grids), batch, 0)
ctx.set_grid(grids); ctx.set_backward_waves(nw)
kkt = pr.make_kkt_batch_tiled(ctx.L, grids, batch, unique=4)
ctx.upload(BUF_KKT, kkt)
capi.debug_profile(ctx)
ctx.riccati_backward(); ctx.sync()
ctx.riccati_backward(); ctx.sync()
p = capi.debug_profile(ctx).astype(np.int64)
print("waves=%d batch=%d kernel %.3f ms" % (nw, batch, ctx.time_phase(0, 3)))
for st in (30, 20, 10):
    row = p[st]
    t0 = row[0]
    print("stage %d type %d dims %d; next stage starts at +%d" % (st, grids[st].type, grids[st].dims, p[st - 1][0] - t0))
    print("  M:", " ".join("%d:%d" % (k, row[k] - t0) for k in range(0, 14) if row[k]))
    print("  V:", " ".join("%d:%d" % (k, row[k] - t0) for k in range(14, 32) if row[k]))
