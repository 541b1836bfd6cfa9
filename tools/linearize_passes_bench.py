"""rtoc_linearize_contact_dynamics against RTOC_OPT_LINEARIZE_DOFS_PER_PASS (tangent directions per pass of the walk: LDS per wave
against passes over the bodies).  iCub jump (N = 30, both soles: surface contacts, lift, flight, impact) or the ANYmal trot.
Usage: linearize_passes_bench.py icub32|icub|anymal [batch] [dofs per pass ...]   (0 = the library's choice)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robotoc_amd import capi, problems as pr, robot_model as rm
from robotoc_amd.types import BUF_CDD, BUF_KKT, BUF_SOL, OPT_LINEARIZE_DOFS_PER_PASS, Records, icub_dims

robot = sys.argv[1] if len(sys.argv) > 1 else "icub32"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sweep = [int(x) for x in sys.argv[3:]] or [0]
m = rm.load_named(robot)
rng = np.random.default_rng(0)
if robot == "anymal":
    dims, grids, _ = pr.config_anymal_trot()
    masks, flip = [], False
    for g in grids:
        masks.append(0b1111 if g.dimf == 12 else 0 if g.dimf == 0 else (0b0110 if flip else 0b1001))
        flip = flip != (g.dimf == 6)
    masks, pos, rot = np.array(masks, dtype=np.uint32), np.zeros((len(grids), 4, 3)), None
else:
    from robotoc_amd.grid import ContactSequence, Event, contact_masks, discretize
    dims = icub_dims(m.nv)
    grids = discretize(30, 0.6, 0.0, ContactSequence([12, 0, 12], [Event("lift", 0.25), Event("impact", 0.36, impact_dimf=12)]))
    masks = contact_masks(grids, [0b11, 0, 0b11], [0b11])
    pos, rot = np.zeros((len(grids), 2, 3)), np.tile(np.eye(3).reshape(9)[None, None], (len(grids), 2, 1))
n = len(grids)
ctx = capi.Context(dims, n, batch, 0)
ctx.set_grid(grids)
ctx.set_robot_model(m)
if rot is None:
    ctx.set_contact_schedule(masks, pos)
else:
    ctx.set_contact_schedule(masks, pos, rot)
S = Records(ctx.L, "sol")
uniq = min(batch, 32)
one = S.zeros(uniq, n)
for b in range(uniq):
    for i in range(n):
        q, v, a = rm.random_configuration(m, rng, 0.8)
        S.f(one, "q")[b, i, :m.nq], S.f(one, "v")[b, i, :m.nv], S.f(one, "a")[b, i, :m.nv] = q, v, a
S.f(one, "f")[...] = rng.uniform(-5, 5, S.f(one, "f").shape)
S.f(one, "beta")[...] = rng.uniform(-1, 1, S.f(one, "beta").shape)
S.f(one, "mu")[...] = rng.uniform(-1, 1, S.f(one, "mu").shape)
ctx.upload(BUF_SOL, np.ascontiguousarray(np.tile(one, (batch // uniq + 1, 1, 1))[:batch]))
ref = None
for dpp in sweep:
    ctx.set_linearize_dofs_per_pass(dpp)
    ctx.upload(BUF_KKT, np.zeros(ctx.shape("kkt")))
    ctx.upload(BUF_CDD, np.zeros(ctx.shape("cdd")))
    ctx.linearize_contact_dynamics(1)
    ctx.sync()
    out = (ctx.download_records(BUF_CDD, "cdd")[:uniq].copy(), ctx.download_records(BUF_KKT, "kkt")[:uniq].copy())
    if ref is None:
        ref = out
    dev = max(float(np.abs(a - b).max()) for a, b in zip(out, ref))
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.linearize_contact_dynamics(1)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    print("%s dofs per pass %2d (asked %2d): %.3f ms / %d x %d grid points = %.1f ns per grid point; max |difference to the first setting| %.2e"
          % (robot, ctx.get_option(OPT_LINEARIZE_DOFS_PER_PASS), dpp, dt * 1e3, batch, n - 1, dt * 1e9 / (batch * (n - 1)), dev))
assert (ctx.status() == 0).all()
ctx.close()
