"""What shader clock does the device sustain under the register kernels?  The PROF build stamps s_memtime (shader-clock cycles) at the
stage tops of instance 0; on a grid without events the whole backward recursion is ONE launch, and a batch that fills the device exactly
once runs it as one round: cycles between the first stage top and the end of the last stage / the launch's duration = the clock.
The rooflines of DESIGN.md are quoted at the 2.4 GHz peak clock.
Usage: RTOC_HIP_LIB=robotoc_amd/librtoc_hip_prof.so python tools/sustained_clock_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.grid import uniform_grid
from robotoc_amd.types import BUF_KKT, icub_dims

cases = [("anymal (riccati_backward_rv_kernel, 8 instances per CU)", pr.config_anymal_trot()[0], 12, 2048, 16),
         ("anymal (riccati_backward_rv_kernel, 4 instances per CU: one wave per SIMD)", pr.config_anymal_trot()[0], 12, 1024, 16),
         ("iCub nv = 32 (riccati_backward_rw_kernel, 4 per CU)", icub_dims(32), 12, 1024, 13),
         ("iCub nv = 35 (riccati_backward_rw2_kernel, 2 per CU)", icub_dims(35), 12, 512, 13)]
for name, dims, dimf, batch, end_slot in cases:
    grids = uniform_grid(30, 0.02, dimf=dimf)
    ctx = capi.Context(dims, len(grids), batch, 0)
    ctx.set_grid(grids)
    ctx.set_backward_register(2)
    ctx.set_fxx_structure(2)
    ctx.upload(BUF_KKT, pr.make_kkt_batch_tiled(ctx.L, grids, batch, unique=4))
    capi.debug_profile(ctx)
    for _ in range(3):
        ctx.riccati_backward(); ctx.sync()
    p = capi.debug_profile(ctx).astype(np.int64)
    ms = min(ctx.time_phase(0, 3) for _ in range(5))
    n = len(grids) - 1                      # stages the kernel runs: n - 1 .. 0
    first, last_end = int(p[n - 1][0]), int(p[0][end_slot])
    stages = [int(p[st - 1][0] - p[st][0]) for st in range(n - 1, 0, -1)]
    print("%s: batch %d, %d stages in one launch of %.3f ms; instance 0: %d cycles from its first stage top to the end of its last stage "
          "(median stage %d) -> %.2f GHz sustained (status nonzero: %d)" % (
              name, batch, n, ms, last_end - first, int(np.median(stages)), (last_end - first) / (ms * 1e6), int((ctx.status() != 0).sum())))
    ctx.close()
