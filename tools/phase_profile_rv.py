"""Cycle stamps of the register-resident backward kernel (PROF build), instance 0: slot k relative to the stage start.
0 stage top | 1 vmcnt(0) passed | 2 Qxx loads issued | 3 PB, G done | 4 P stores issued | 5 Cholesky + inverse done |
6, 8, 10 W[:, t] done | 7, 9, 11 F[:, t] done | 12 DMA of the next record issued | 13 policy products + K stores | 14 Z Z^T |
15 transposes | 16 end.      Usage: RTOC_HIP_LIB=robotoc_amd/librtoc_hip_prof.so python tools/phase_profile_rv.py [batch] [trot|icub32|icub35]
Register-wide kernel of the iCub shapes (riccati_backward_rw.hpp): 0 stage top | 1 Bv, Quu loads issued + everything landed | 2 z |
3 PB, G | 4 Cholesky + inverse | 5 t, k | 6 H, transposes | 7 Z^T | 8 K | 9 Qxx + Z Z^T | 10 s, strip DMA | 11 W / F column loop |
12 DMA of A issued | 13 transposes, end."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = sys.argv[2] if len(sys.argv) > 2 else "trot"
dims, grids, _ = {"trot": pr.config_anymal_trot, "icub32": lambda: pr.config_icub_jump(nv=32), "icub35": lambda: pr.config_icub_jump(nv=35)}[cfg]()
ctx = capi.Context(dims, len(grids), batch, 0)
ctx.set_grid(grids)
ctx.set_backward_register(2)
kkt = pr.make_kkt_batch_tiled(ctx.L, grids, batch, unique=4)
ctx.upload(BUF_KKT, kkt)
capi.debug_profile(ctx)
ctx.riccati_backward(); ctx.sync()
ctx.riccati_backward(); ctx.sync()
p = capi.debug_profile(ctx).astype(np.int64)
print("batch=%d kernel sequence %.3f ms" % (batch, ctx.time_phase(0, 3)))
for st in ((40, 30, 25, 20, 10) if cfg == "trot" else (31, 28, 22, 5)):
    row = p[st]
    t0 = row[0]
    print("stage %d type %d; next stage top at +%d" % (st, grids[st].type, p[st - 1][0] - t0))
    print("   ", " ".join("%d:%d" % (k, row[k] - t0) for k in range(0, 17) if row[k]))
    if cfg == "icub35" and row[16]:   # the second wave of the instance (riccati_backward_rw2.hpp), relative to the first wave's stage top
        print("    wave 1:", " ".join("%d:%d" % (k - 16, row[k] - t0) for k in range(16, 32) if row[k]))
