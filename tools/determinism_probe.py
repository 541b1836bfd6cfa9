"""Run the headline sweep (4096 distinct ANYmal trot instances) repeatedly on one context and compare every downloaded
record with the first run's, bit for bit: a race in the backward kernel shows in the Riccati records, one in the forward
kernel in the direction records only.  Prints instance / stage / field of whatever differs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_KKT, BUF_DX0, BUF_RIC, BUF_DIR, Records

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = os.environ.get("PROBE_CONFIG", "anymal_trot")   # anymal_trot | anymal_jump_sto | icub35 | icub32
batch = int(os.environ.get("PROBE_BATCH", "4096" if cfg.startswith("anymal") else "512"))
dims, grids, _ = {"anymal_trot": pr.config_anymal_trot, "anymal_jump_sto": pr.config_anymal_jump_sto,
                  "icub35": lambda: pr.config_icub_jump(nv=35), "icub32": lambda: pr.config_icub_jump(nv=32)}[cfg]()
ctx = capi.Context(dims, len(grids), batch, 0)
L = ctx.L
ctx.set_grid(grids)
kkt = pr.make_kkt_batch_unique(L, grids, batch)
dx0 = pr.make_dx0_unique(L, batch)
ctx.upload(BUF_KKT, kkt)
ctx.upload(BUF_DX0, dx0)
ref = None
bad = 0
for it in range(reps):
    if len(sys.argv) > 2 and sys.argv[2] == "fwdonly" and it > 0:
        ctx.riccati_forward()
    else:
        ctx.riccati_backward()
        ctx.riccati_forward()
    ric = ctx.download_records(BUF_RIC, "ric").copy()
    d = ctx.download_records(BUF_DIR, "dir").copy()
    if ref is None:
        ref = (ric, d)
        continue
    for name, a, b in (("ric", ric, ref[0]), ("dir", d, ref[1])):
        ne = (a.view(np.uint64) != b.view(np.uint64))
        if ne.any():
            bad += 1
            R = Records(L, name)
            idx = np.argwhere(ne)
            inst = sorted(set(idx[:, 0].tolist()))
            stages = sorted(set(idx[:, 1].tolist()))
            fields = []
            for f in R.names:
                o, n = R.offset(f), int(np.prod(R.shapes[f]))
                if ne[..., o:o + n].any():
                    fields.append(f)
            print("run %d: %s differs in %d words, instances %s%s, stages %s..%s, fields %s" % (
                it, name, int(ne.sum()), inst[:8], "..." if len(inst) > 8 else "", stages[0], stages[-1], fields))
print("%s x %d: %d runs, %d record sets differ from the first run" % (cfg, batch, reps, bad))
ctx.close()
