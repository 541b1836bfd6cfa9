"""Backward / forward kernel times for every BASELINE.json configuration (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import algorithmic_bytes
cfgs = [("iiwa14_unconstr", pr.config_iiwa14, 4096), ("anymal_trot", pr.config_anymal_trot, 4096),
        ("anymal_jump_sto", pr.config_anymal_jump_sto, 4096), ("icub35_jump", lambda: pr.config_icub_jump(nv=35), 1024),
        ("icub32_jump", lambda: pr.config_icub_jump(nv=32), 1024)]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, fn, batch in cfgs:
    if only and only not in name:
        continue
    dims, grids, info = fn()
    ctx = capi.Context(dims, len(grids), batch, 0); L = ctx.L; ctx.set_grid(grids)
    if name.startswith("iiwa"):
        from robotoc_amd.types import Records
        k1 = Records(L, "kkt").zeros(4, len(grids))
        for b in range(4):
            pr.fill_unconstr_instance(L, len(grids), k1[b], np.random.default_rng(b))
        ctx.upload(BUF_KKT, np.ascontiguousarray(np.tile(k1, (batch // 4, 1, 1))))
        ctx.unconstr_backward(info["dt"])  # materialise A, B
    else:
        ctx.upload(BUF_KKT, pr.make_kkt_batch_tiled(L, grids, batch, unique=4))
    ctx.upload(BUF_DX0, np.tile(pr.make_dx0(L, 4), (batch // 4, 1)))
    ctx.time_phase(4, 1)
    mb, mf = ctx.time_phase(0, 3), ctx.time_phase(1, 3)
    bb, bf = algorithmic_bytes(L, grids, batch, "backward"), algorithmic_bytes(L, grids, batch, "forward")
    print("%-16s batch %5d stages %2d: backward %8.3f ms (%6.1f GB/s alg)  forward %7.3f ms (%6.1f GB/s)  -> %9.0f sweeps/s  status!=0: %d"
          % (name, batch, len(grids), mb, bb / mb / 1e6, mf, bf / mf / 1e6, batch / (mb + mf) * 1e3, int((ctx.status() != 0).sum())))
    ctx.close()
