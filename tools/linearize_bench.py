"""Time rtoc_linearize_contact_dynamics at the bench size (ANYmal trot, batch x 46 grid points).  Usage: linearize_bench.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr, robot_model as rm
from robotoc_amd.types import BUF_SOL
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
        one[b, i, o[3]:o[3] + 12] = rng.uniform(-5, 5, 12)
        one[b, i, o[4]:o[4] + 12] = rng.uniform(-20, 20, 12)
ctx.upload(BUF_SOL, np.ascontiguousarray(np.tile(one, (batch // 64 + 1, 1, 1))[:batch]))
for aug in (0, 1):
    if aug:
        ctx.upload(0, np.zeros(ctx.shape("kkt")))
    ctx.linearize_contact_dynamics(aug); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.linearize_contact_dynamics(aug)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    print("linearize_contact_dynamics%s: %.3f ms / %d x %d grid points = %.1f ns per grid point" % (" + multiplier terms" if aug else "", dt * 1e3, batch, len(grids) - 1, dt * 1e9 / (batch * (len(grids) - 1))))
ctx.close()
