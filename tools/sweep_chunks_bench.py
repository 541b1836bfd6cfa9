"""Backward+forward sweep with the forward of chunk i under the backward of chunk i+1 (RTOC_OPT_SWEEP_CHUNKS).  Usage: sweep_chunks_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_DX0, BUF_KKT, OPT_SWEEP_CHUNKS
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dims, grids, _ = pr.config_anymal_trot()
ctx = capi.Context(dims, len(grids), batch, 0)
L = ctx.L
ctx.set_grid(grids)
tile = lambda a: np.ascontiguousarray(np.tile(a, (batch // a.shape[0] + 1,) + (1,) * (a.ndim - 1))[:batch])
ctx.upload(BUF_KKT, tile(pr.make_kkt_batch_unique(L, grids, 64)))
ctx.upload(BUF_DX0, tile(pr.make_dx0_unique(L, 64)))
for nch in (1, 2, 4, 8, 16):
    capi._chk(capi.lib().rtoc_set_option(ctx._h, OPT_SWEEP_CHUNKS, nch))
    ctx.time_phase(4, 2)
    print("chunks %2d: sweep %.3f ms (best of 5 x 3 launches)" % (nch, min(ctx.time_phase(4, 3) for _ in range(5))), "status", int((ctx.status() != 0).sum()))
print("backward %.3f forward %.3f" % (ctx.time_phase(0, 3), ctx.time_phase(1, 3)))
ctx.close()
