"""Per-phase cycle breakdown of the backward kernel for a named config (block 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT
name = sys.argv[1]; batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
fn = {"icub35": lambda: pr.config_icub_jump(nv=35), "icub32": lambda: pr.config_icub_jump(nv=32),
      "anymal": pr.config_anymal_trot, "sto": pr.config_anymal_jump_sto}[name]
dims, grids, _ = fn()
names = ["phase-trans", "regs->LDS", "z + PB + G + lu", "PAa mfma + H", "w=A^T z", "F init+LLT+chain", "solve", "GK mfma", "KtGK + Hk", "sym", "sto", "writes", "end"]
ctx = capi.Context(dims, len(grids), batch, 0); ctx.set_grid(grids); L = ctx.L
if len(sys.argv) > 3: ctx.set_backward_waves(int(sys.argv[3]))
ctx.upload(BUF_KKT, pr.make_kkt_batch_tiled(L, grids, batch, unique=2))
capi.debug_profile(ctx); ctx.riccati_backward(); ctx.sync(); ctx.riccati_backward(); ctx.sync()
p = capi.debug_profile(ctx); ms = ctx.time_phase(0, 2)
print("%s batch=%d kernel %.3f ms" % (name, batch, ms))
for st in (len(grids) - 3, 5):
    row = p[st]; d = np.diff(row[:13])
    print(" stage %2d (type %d dims %d sto %d) total %7d: " % (st, grids[st].type, grids[st].dims, grids[st].sto, row[12] - row[0]) + " | ".join("%s %d" % (n, x) for n, x in zip(names, d)))
    print("    tile-split kernel: F accumulators loaded at +%d (from phase 5 start), LLT + inverse by wave 0 took %d, F chain %d" % (row[13] - row[5], row[14] - row[13], row[6] - row[14]))
    print("    solve phase: barrier passed at +%d, prefetch issued +%d, policy products of this wave %d, wait for the others %d" % (row[15] - row[6], row[16] - row[15], row[17] - row[16], row[7] - row[17]))
    print("    matrix wave F done at +%d, vector wave solve done at +%d (B4 at +%d); vector wave interval-4 done at +%d (B5 at +%d)" % (row[13] - row[0], row[14] - row[0], row[6] - row[0], row[15] - row[0], row[10] - row[0]))
    v = row[16:30] - row[0]
    vn = ["stage top", "pt done", "B1 passed", "z done", "lu' done (wait G)", "G flag seen", "LLT+Ginv done", "w done (wait H)", "H flag seen", "Kt mfma done", "B4 passed", "prefetch issued", "B5 passed", "stage end"]
    print("    vector wave: " + " | ".join("%s +%d" % (n, x) for n, x in zip(vn, v)))
