"""Forward-kernel timing of the library named by RTOC_HIP_LIB (tuning runs: variants built with other FWD_HEAD / FWD_REST_PARTS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0

batch = 4096
dims, grids, _ = pr.config_anymal_trot()
ctx = capi.Context(dims, len(grids), batch, 0)
ctx.set_grid(grids)
ctx.upload(BUF_KKT, pr.make_kkt_batch_tiled(ctx.L, grids, batch, unique=8))
ctx.upload(BUF_DX0, np.tile(pr.make_dx0(ctx.L, 8), (batch // 8 + 1, 1))[:batch])
ctx.time_phase(0, 3)
ctx.time_phase(1, 3)
ms = [ctx.time_phase(1, 10) for _ in range(3)]
sw = [ctx.time_phase(4, 10) for _ in range(2)]
print("%-40s forward %.3f %.3f %.3f ms   sweep %.3f %.3f ms   status %d" % (os.path.basename(os.environ.get("RTOC_HIP_LIB", "default")),
      ms[0], ms[1], ms[2], sw[0], sw[1], int((ctx.status() != 0).sum())))
ctx.close()
