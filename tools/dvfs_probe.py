"""Why does riccati_backward_rs4_kernel take 2.3 ms in the sweep loop and 2.55 ms inside the closed-loop iteration?
Same kernel, same grid, same records: timed (a) back to back with itself, (b) each launch right behind the rigid-body
linearisation (9 ms of dense fp64 VALU work), (c) right behind the condensation, (d) behind an idle gap.
Usage: dvfs_probe.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr, robot_model as rm
from robotoc_amd.types import BUF_SOL, BUF_KKT, BUF_DX0
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = rm.load_named("anymal")
dims, grids, _ = pr.config_anymal_trot()
ctx = capi.Context(dims, len(grids), batch, 0)
ctx.set_grid(grids)
ctx.set_robot_model(m)
masks, flip = [], False
for g in grids:
    masks.append(0b1111 if g.dimf == 12 else 0 if g.dimf == 0 else (0b0110 if flip else 0b1001))
    flip = flip != (g.dimf == 6)
ctx.set_contact_schedule(np.array(masks, dtype=np.uint32), np.zeros((len(grids), 4, 3)))
rng = np.random.default_rng(0)
L = ctx.L
o = L.sol.off
one = np.zeros((64, len(grids), L.sol.stride))
for b in range(64):
    for i in range(len(grids)):
        q, v, a = rm.random_configuration(m, rng, 0.8)
        one[b, i, o[0]:o[0] + m.nq], one[b, i, o[1]:o[1] + m.nv], one[b, i, o[2]:o[2] + m.nv] = q, v, a
ctx.upload(BUF_SOL, np.ascontiguousarray(np.tile(one, (batch // 64 + 1, 1, 1))[:batch]))
ctx.upload(BUF_KKT, pr.make_kkt_batch_tiled(L, grids, batch, unique=8))
ctx.upload(BUF_DX0, pr.make_dx0(L, batch))
ctx.time_phase(0, 3)
res = {}
res["back to back"] = ctx.time_phase(0, 20)   # time_phase returns the mean per repetition
t = []
for _ in range(10):
    ctx.time_phase(7, 1)           # linearisation, synchronised
    t.append(ctx.time_phase(0, 1))
res["each launch right behind the linearisation"] = float(np.mean(t))
t = []
for _ in range(10):
    ctx.time_phase(7, 3)
    t.append(ctx.time_phase(0, 1))
res["behind three linearisations"] = float(np.mean(t))
t = []
for _ in range(10):
    time.sleep(0.05)
    t.append(ctx.time_phase(0, 1))
res["behind a 50 ms idle gap"] = float(np.mean(t))
t = []
for _ in range(10):
    ctx.time_phase(1, 3)           # forward recursions (bandwidth-bound)
    t.append(ctx.time_phase(0, 1))
res["behind three forward recursions"] = float(np.mean(t))
res["back to back again"] = ctx.time_phase(0, 20)
for k, v in res.items():
    print("backward %-45s %.3f ms" % (k, v))
