"""numpy restatement of how the reference EVALUATES its inequality rows -- shared by the CPU test that pins it to the
reference's own sources (tests/test_constraints_vs_reference.py, oracle/_ref) and by the GPU tests that compare the device
kernels with it (tests/test_contact_constraints.py).  Test infrastructure.

Joint limits (src/constraints/joint_position_lower_limit.cpp:40-77 and its five siblings), in the row order of
robotoc_amd.types.joint_limit_rows: g = sign z - bound.  Friction cone (src/constraints/friction_cone.cpp:100-191,
friction_cone.hpp:102-120).  pdipm.hxx:12-31 for slack / dual / complementarity."""
import numpy as np


def init_slack_dual(g, barrier):
    """setSlack + setSlackAndDualPositive: slack = -g clipped at sqrt(barrier), dual = barrier / slack"""
    slack = np.maximum(-np.asarray(g, dtype=float), np.sqrt(barrier))
    return slack, barrier / slack


def joint_limit_values(rows, bounds, q, v, u, floating):
    """g of every row; q carries the quaternion of a free-flyer (nq = nv + 1): joint entry j sits at j + 1"""
    off = 1 if floating else 0
    g = np.zeros(len(rows))
    for r, w in enumerate(rows):
        z = q[w.index + off] if w.var == 0 else (v[w.index] if w.var == 1 else u[w.index])
        g[r] = w.sign * z - bounds[r]
    return g


def joint_limit_active(rows, time_stage, impact):
    """ConstraintsData's stage mask (constraints_data.cpp:20-45): a row acts iff time_stage >= its level, never on impact grids"""
    return np.array([(not impact) and time_stage >= w.level for w in rows])


def joint_limit_gradients(rows, active, dual, nv, nu):
    """evalDerivatives: l += sign * dual on the entry the row acts on"""
    lx, lu = np.zeros(2 * nv), np.zeros(nu)
    for r, w in enumerate(rows):
        if not active[r]:
            continue
        if w.var == 2:
            lu[w.index] += w.sign * dual[r]
        else:
            lx[w.index + (nv if w.var == 1 else 0)] += w.sign * dual[r]
    return lx, lu


def cone_world(mu):
    m = mu / np.sqrt(2.0)
    return np.array([[0, 0, -1], [1, 0, -m], [-1, 0, -m], [0, 1, -m], [0, -1, -m]], dtype=float)


def friction_cone_rows(mu, R_surface, R_wf, w_world, f_local, exact_jacobian):
    """One active contact: g (5), dg/dq (5 x nv), dg/df (5 x 3).
    w_world: 3 x nv, the world-aligned angular Jacobian of the contact frame.  The reference takes the LOCAL-frame angular
    Jacobian (pinocchio::getFrameJacobian(..., LOCAL, ...)) and crosses it with the WORLD-frame force (robot.hxx:247-287);
    exact_jacobian: w_world x f_W, the derivative of R_wf(q) f."""
    cone_local = cone_world(mu) @ R_surface.T
    fW = R_wf @ f_local[:3]
    g = cone_local @ fW
    w = w_world if exact_jacobian else R_wf.T @ w_world
    dfW_dq = np.cross(w.T, fW).T          # column j: w_j x f_W
    return g, cone_local @ dfW_dq, cone_local @ R_wf
