"""Single-instance latency of the backward recursion: serial chain vs horizon scan (HIP events).
usage: python tools/scan_latency.py [anymal|icub32|icub35] [batch] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robotoc_amd import capi, problems as pr  # noqa: E402
from robotoc_amd.types import BUF_DX0, BUF_KKT  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "anymal"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
from robotoc_amd import grid as G  # noqa: E402
from robotoc_amd.types import anymal_dims, icub_dims  # noqa: E402
cfg = {"anymal_plain": lambda: (anymal_dims(), G.uniform_grid(46, 0.02, 12), None),
       "icub32_plain": lambda: (icub_dims(32), G.uniform_grid(33, 0.02, 12), None),
       "anymal": pr.config_anymal_trot, "icub32": lambda: pr.config_icub_jump(nv=32),
       "icub35": lambda: pr.config_icub_jump(nv=35)}[name]
dims, grids, _ = cfg()
for g in grids:
    g.sto = 0
    g.sto_next = 0
for scan in (False, True):
    ctx = capi.Context(dims, len(grids), batch, 0)
    ctx.set_grid(grids)
    ctx.set_backward_scan(scan)
    ctx.upload(BUF_KKT, pr.make_kkt_batch_tiled(ctx.L, grids, batch, unique=min(batch, 4)))
    ctx.upload(BUF_DX0, pr.make_dx0(ctx.L, batch))
    ctx.time_phase(4, 5)
    print("%s batch %d %s: backward %.4f ms, forward %.4f ms, sweep %.4f ms" % (
        name, batch, "scan" if scan else "serial", ctx.time_phase(0, reps), ctx.time_phase(1, reps), ctx.time_phase(4, reps)))
    ctx.close()
