"""Quick per-kernel timing (HIP events on the launch stream) for tuning runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0

def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dims, grids, _ = pr.config_anymal_trot()
    ctx = capi.Context(dims, len(grids), batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    kkt = pr.make_kkt_batch_tiled(L, grids, batch, unique=8)
    dx0 = np.tile(pr.make_dx0(L, 8), (batch // 8 + 1, 1))[:batch]
    ctx.upload(BUF_KKT, kkt); ctx.upload(BUF_DX0, dx0)
    nst = len(grids)
    only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for nw in ((only,) if only else (1, 2, 3)):
        ctx.set_backward_waves(nw)
        ctx.time_phase(0, 2)
        ms = ctx.time_phase(0, 5)
        print("backward NW=%d batch=%d stages=%d: %.3f ms/launch  -> %.1f sweeps(bwd)/s" % (nw, batch, nst, ms, batch / ms * 1e3))
    ctx.time_phase(1, 2)
    ms = ctx.time_phase(1, 5)
    print("forward: %.3f ms/launch" % ms)
    for ch in (() if "nosweep" in sys.argv else (1, 4)):
        ctx.set_sweep_chunks(ch)
        ctx.time_phase(4, 2)
        ms = ctx.time_phase(4, 5)
        print("sweep (backward+forward) chunks=%2d: %.3f ms  -> %.1f sweeps/s" % (ch, ms, batch / ms * 1e3))
    print("status nonzero:", int((ctx.status() != 0).sum()))
    ctx.close()

if __name__ == "__main__":
    main()
