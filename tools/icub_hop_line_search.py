"""iCub through stand - flight - touch-down WITH its inequality rows (joint limits + friction or wrench cones): Gauss-Newton
iterations stall without a line search (round 2); with the filter line search on the device they should converge.
Usage: icub_hop_line_search.py"""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "tests"))
import numpy as np

from robotoc_amd import capi, robot_model as rm
from robotoc_amd.grid import ContactSequence, Event, contact_masks, discretize
from robotoc_amd.types import BUF_SOL, GRID_IMPACT, Records, icub_dims, joint_limit_rows
from test_contact_constraints import Q_ICUB, limits

m = rm.load_named("icub")
nv, nq, nu = m.nv, m.nq, m.nv - 6
for cones in ("friction", "wrench"):
    for ls in ((False, True) if not os.environ.get("HOP_NO_LS") else (False,)):
        cone_rows = 10 if cones == "friction" else 34
        dims = icub_dims(nv, nc_max=(6 * nu + cone_rows + 7) & ~7)
        cs = ContactSequence([12, 0, 12], [Event("lift", 0.25), Event("impact", 0.36, impact_dimf=12)])
        grids = discretize(30, 0.6, 0.0, cs)
        n = len(grids)
        masks = contact_masks(grids, [0b11, 0, 0b11], [0b11])
        place = [m.frame_placement(Q_ICUB, c) for c in range(2)]
        pos = np.tile(np.array([p for _, p in place])[None], (n, 1, 1))
        rot = np.tile(np.array([R.reshape(9) for R, _ in place])[None], (n, 1, 1))
        ctx = capi.Context(dims, n, 1, 0)
        ctx.set_grid(grids)
        ctx.set_robot_model(m)
        ctx.set_contact_schedule(masks, pos, rot)
        ctx.set_constraint_rows(joint_limit_rows(dims))
        if cones == "friction":
            ctx.set_friction_cones(2, 6)
        else:
            ctx.set_wrench_cones(2)
        ctx.set_impact_cones(False)
        LIM = [float(x) for x in os.environ.get("HOP_LIMITS", "2.5,8.0,120.0").split(",")]
        ctx.set_constraint_bounds(limits(nu, *LIM), float(os.environ.get("HOP_BARRIER", "1e-3")), 0.995)
        if cones == "friction":
            ctx.set_friction_coefficients(np.full(2, 0.6))
        else:
            ctx.set_wrench_cone_params(np.array([[0.2, 0.1, 0.9]] * 2))
        wq = np.concatenate([np.full(6, 10.0), np.full(nu, 0.1)])
        ctx.set_configuration_cost(Q_ICUB, np.zeros(nv), np.zeros(nu), wq, np.full(nv, 0.1), np.full(nv, 1e-3), np.full(nu, 1e-4), 10 * wq,
                                   np.full(nv, 0.1), q_weight_impact=wq, v_weight_impact=np.full(nv, 0.1), dv_weight_impact=np.full(nv, 1e-3))
        ctx.set_initial_state(np.concatenate([Q_ICUB, np.zeros(nv)])[None])
        S = Records(ctx.L, "sol")
        sol = S.zeros(1, n)
        mass = sum(m.mass[i] for i in range(m.njoints))
        f0 = np.concatenate([np.concatenate([R.T @ np.array([0, 0, 9.81 * mass / 2]), np.zeros(3)]) for R, _ in place])
        S.f(sol[0], "q")[:, :nq] = Q_ICUB
        for i in range(n):
            if masks[i] and grids[i].type != GRID_IMPACT:
                S.f(sol[0, i], "f")[:12] = f0
        ctx.upload(BUF_SOL, sol)
        ctx.contact_init_constraints()
        if ls:
            ctx.set_line_search(True)
            ctx.line_search_clear()
        hist, steps = [], []
        for it in range(300):
            e = ctx.contact_update_solution(0.995)[0]
            hist.append(e)
            steps.append(ctx.download(6, (1, 2))[0, 0])
            if e < 1e-8 or not np.isfinite(e):
                break
        print(cones, "line search", ls, "iters", len(hist), "status", ctx.status(), ["%.1e" % e for e in hist[:5]], "...",
              ["%.1e" % e for e in hist[-4:]], "steps", ["%.2f" % s for s in steps[:12]], "...", ["%.2f" % s for s in steps[-4:]])
        ctx.close()
