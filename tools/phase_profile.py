"""Per-phase cycle breakdown of the backward kernel (block 0), s_memtime stamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims, grids, _ = pr.config_anymal_trot()
names = ["phase-trans", "g2s copies", "z + PB + G + lu", "PAa mfma + H", "w=A^T z", "F init+chain mfma", "LLT+solve", "GK mfma", "KtGK mfma + Hk", "F->sP, sym", "sto scal", "writes", "end"]
for nw in (1, 2, 3):
    ctx = capi.Context(dims, len(grids), batch, 0)
    ctx.set_grid(grids); ctx.set_backward_waves(nw)
    L = ctx.L
    kkt = pr.make_kkt_batch_tiled(L, grids, batch, unique=4)
    ctx.upload(BUF_KKT, kkt)
    capi.debug_profile(ctx)
    ctx.riccati_backward(); ctx.sync()
    ctx.riccati_backward(); ctx.sync()
    p = capi.debug_profile(ctx)
    ms = ctx.time_phase(0, 3)
    print("NW=%d batch=%d kernel %.3f ms" % (nw, batch, ms))
    for st in (40, 30, 20, 16, 15, 10):
        row = p[st]
        d = np.diff(row[:13])
        tot = row[12] - row[0]
        print(" stage %2d (type %d dims %d) total %7d ticks: " % (st, grids[st].type, grids[st].dims, tot) + " | ".join("%s %d" % (n, x) for n, x in zip(names, d)))
    nxt = p[20, 0] - p[21, 0]
    print(" stage-to-stage ticks (21->20):", nxt)
    ctx.close()
