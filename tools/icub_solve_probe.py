#!/usr/bin/env python3
"""Does the iCub hop (BASELINE configs[3]'s robot: stand - flight - touch-down of both soles, N = 30) with its inequality rows converge on
the device, and with which line search?  KKT histories of OCPSolver::updateSolution from the standing guess: joint limits + friction
cones on the soles (what examples/icub/python/jump_sto.py:55-67 poses) or the wrench cones, no line search / filter / merit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from robotoc_amd import capi, robot_model as rm
from robotoc_amd.grid import ICUB_Q_STANDING, ContactSequence, Event, discretize, contact_masks
from robotoc_amd.types import BUF_SOL, GRID_IMPACT, Records, icub_dims, joint_limit_rows

nv_sel = int(sys.argv[1]) if len(sys.argv) > 1 else 35
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
m = rm.load_named("icub32" if nv_sel == 32 else "icub")
nv, nq, nu = m.nv, m.nq, m.nv - 6
qs = np.array(ICUB_Q_STANDING, dtype=float)
if nv == 32:
    qs = np.delete(qs, [19, 20, 21])
variants = [(c, l, ms) for c in ("friction", "wrench") for l, ms in (("off", 0.05), ("filter", 0.05), ("merit", 0.05), ("merit", 0.005))]
if len(sys.argv) > 3:
    variants = [v for v in variants if v[1] in sys.argv[3].split(",")]
for cones, ls, min_step in variants:
    if True:
        cone_rows = 10 if cones == "friction" else 34
        dims = icub_dims(nv, nc_max=(6 * nu + cone_rows + 7) & ~7)
        grids = discretize(30, 0.6, 0.0, ContactSequence([12, 0, 12], [Event("lift", 0.25), Event("impact", 0.36, impact_dimf=12)]))
        n = len(grids)
        masks = contact_masks(grids, [0b11, 0, 0b11], [0b11])
        place = [m.frame_placement(qs, c) for c in range(2)]
        pos = np.tile(np.array([p for _, p in place])[None], (n, 1, 1))
        rot = np.tile(np.array([R.reshape(9) for R, _ in place])[None], (n, 1, 1))
        ctx = capi.Context(dims, n, 1, 0)
        ctx.set_grid(grids); ctx.set_robot_model(m); ctx.set_contact_schedule(masks, pos, rot)
        ctx.set_constraint_rows(joint_limit_rows(dims))
        if cones == "friction":
            ctx.set_friction_cones(2, 6)
        else:
            ctx.set_wrench_cones(2)
        ctx.set_constraint_bounds(np.concatenate([np.full(2 * nu, 2.5), np.full(2 * nu, 5.0), np.full(2 * nu, 60.0)]), 1.0e-3, 0.995)
        if cones == "friction":
            ctx.set_friction_coefficients(np.full(2, 0.6))
        else:
            ctx.set_wrench_cone_params(np.array([[0.1, 0.05, 0.6]] * 2))
        wq = np.concatenate([np.full(6, 10.0), np.full(nu, 0.1)])
        ctx.set_configuration_cost(qs, np.zeros(nv), np.zeros(nu), wq, np.full(nv, 0.1), np.full(nv, 1e-3), np.full(nu, 1e-4), 10 * wq, np.full(nv, 0.1),
                                   q_weight_impact=wq, v_weight_impact=np.full(nv, 0.1), dv_weight_impact=np.full(nv, 1e-3))
        ctx.set_initial_state(np.concatenate([qs, np.zeros(nv)])[None])
        if ls != "off":
            ctx.set_line_search(True, 0.75, min_step)
            ctx.set_line_search_method(ls)
            ctx.line_search_clear()
        S = Records(ctx.L, "sol")
        sol = S.zeros(1, n)
        mass = sum(m.mass[i] for i in range(m.njoints))
        f0 = np.concatenate([np.concatenate([R.T @ np.array([0, 0, 9.81 * mass / 2]), np.zeros(3)]) for R, _ in place])
        S.f(sol[0], "q")[:, :nq] = qs
        for i in range(n):
            if masks[i] and grids[i].type != GRID_IMPACT:
                S.f(sol[0, i], "f")[:12] = f0
        ctx.upload(BUF_SOL, sol)
        ctx.contact_init_constraints()
        hist = []
        for it in range(iters):
            e = ctx.contact_update_solution(0.995)[0]
            hist.append(e)
            if e < 1e-7 or not np.isfinite(e):
                break
        print("nv %d %-8s line search %-6s (min step %.3f): %3d iterations, status %s, KKT %s ... %s" % (
            nv, cones, ls, min_step, len(hist), ctx.status(), ["%.1e" % e for e in hist[:4]], ["%.1e" % e for e in hist[-5:]]), flush=True)
        ctx.close()
