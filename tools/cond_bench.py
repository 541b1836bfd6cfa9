#!/usr/bin/env python3
"""rtoc_condense of the headline batch (4096 distinct ANYmal trot instances, 72 joint-limit rows, 4 friction cones) with the
role-split kernel and with the register-chained kernel (RTOC_OPT_CONDENSE_REGISTER), timed with events on the context's stream,
interleaved (the second timing in a process runs at a higher clock); every record the two leave behind compared on the device."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_KKT, joint_limit_rows

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with_rows = "norows" not in sys.argv
dims, grids, _ = pr.config_anymal_trot()
n = len(grids)
dev = "cuda:0"
ctx = capi.Context(dims, n, batch, 0)
L = ctx.L
ctx.set_grid(grids)
z = lambda w: torch.zeros((batch, n, getattr(L, w).stride), dtype=torch.float64, device=dev)
if with_rows:
    ctx.set_constraint_rows(joint_limit_rows(dims))
    ctx.set_friction_cones(4, 3)
kkt0, cdd0 = pr.make_precondense_batch_unique(L, grids, batch, seed=0, backend="torch", device=dev, out=(z("kkt"), z("cdd")))
con0 = pr.make_constraint_batch_unique(L, grids, batch, seed=0, backend="torch", device=dev, out=z("con"))
cone_t = pr.make_cone_batch_unique(L, grids, batch, 4, seed=0, backend="torch", device=dev).contiguous()
kkt_w, cdd_w, con_w = torch.empty_like(kkt0), torch.empty_like(cdd0), torch.empty_like(con0)
for b_, t_ in ((BUF_KKT, kkt_w), (BUF_CDD, cdd_w), (BUF_CON, con_w), (BUF_CONE, cone_t)):
    ctx.bind(b_, t_.data_ptr())


def restore():
    kkt_w.copy_(kkt0)
    cdd_w.copy_(cdd0)
    con_w.copy_(con0)
    torch.cuda.synchronize()


res, times = {}, {}
for name, on in (("role-split", False), ("register", True)) * 3:
    ctx.set_condense_register(bool(on))
    ctx.clear_status()
    t = []
    for _ in range(4):
        restore()
        t.append(ctx.time_phase(2, 1))
    bad = int((ctx.status() != 0).sum())
    res[name] = (kkt_w.clone(), cdd_w.clone(), con_w.clone())
    times.setdefault(name, []).extend(t[1:])
    print("%-11s rtoc_condense ms: min %.3f median %.3f  status != 0: %d" % (name, min(t[1:]), sorted(t[1:])[1], bad), flush=True)
for name, t in times.items():
    print("%-11s over the interleaved runs: min %.3f  median %.3f" % (name, min(t), sorted(t)[len(t) // 2]))
for w, (a, b), lay in zip(("kkt", "cdd", "con"), zip(res["role-split"], res["register"]), (L.kkt, L.cdd, L.con)):
    worst = []
    for f in range(lay.nfields):
        lo = lay.off[f]
        hi = lay.off[f + 1] if f + 1 < lay.nfields else lay.stride
        if hi <= lo:
            continue
        x, y = a[:, :n - 1, lo:hi], b[:, :n - 1, lo:hi]
        den = x.abs().amax(dim=2, keepdim=True).clamp_min(1.0)
        e = ((x - y).abs() / den)
        e = torch.nan_to_num(e, nan=1e300)
        m = float(e.max())
        if m > 1e-11:
            idx = (e == e.max()).nonzero()[0]
            worst.append((f, m, tuple(int(v) for v in idx)))
    print(w, "fields that differ by more than 1e-11 (field, error, (instance, stage, entry)):", worst if worst else "none")
ctx.close()
