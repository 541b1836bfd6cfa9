"""Counter-collection driver (rocprofv3 --pmc ... -- python tools/pmc_driver.py): launches every hot-path kernel
DESIGN.md quotes at its bench.py size, through the C ABI only -- no torch in the process (rocprofv3's counter
collection segfaults inside torch's first copy kernel on this image; profiles/r02_pmc_notes.txt).  Stage data:
64 distinct instances tiled to the batch (counter values do not depend on the data values: no data-dependent
branches in the kernels).  Usage: pmc_driver.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_DX0, BUF_KKT, icub_dims, joint_limit_rows


def tile(a, batch):
    reps = (batch + a.shape[0] - 1) // a.shape[0]
    return np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:batch])


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    uniq = 64
    # ---- ANYmal trot, 4096 instances: sweep + SQP hot path ----
    dims, grids, _ = pr.config_anymal_trot()
    batch = 4096
    ctx = capi.Context(dims, len(grids), batch, 0)
    L = ctx.L
    ctx.set_grid(grids)
    ctx.upload(BUF_KKT, tile(pr.make_kkt_batch_unique(L, grids, uniq), batch))
    ctx.upload(BUF_DX0, tile(pr.make_dx0_unique(L, uniq), batch))
    for _ in range(reps):
        ctx.riccati_backward()
        ctx.riccati_forward()
    assert (ctx.status() == 0).all()
    rows = joint_limit_rows(dims)
    ctx.set_constraint_rows(rows)
    ctx.set_friction_cones(4, 3)
    kkt, cdd = pr.make_precondense_batch_unique(L, grids, uniq)
    kkt, cdd = tile(kkt, batch), tile(cdd, batch)
    con = tile(pr.make_constraint_batch_unique(L, grids, uniq), batch)
    ctx.upload(BUF_CONE, tile(pr.make_cone_batch_unique(L, grids, uniq, 4), batch))
    for _ in range(reps):
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_CDD, cdd)
        ctx.upload(BUF_CON, con)
        ctx.condense()
        ctx.riccati_backward()
        ctx.riccati_forward()
        ctx.expand(0.995)
        ctx.update()
    assert (ctx.status() == 0).all()
    # the condensation pipeline that is not this shape's default (RTOC_OPT_CONDENSE_SPLIT: 0 = one kernel with the cone rows inside,
    # 1 = mjtjinv_kernel + condense_kernel)
    from robotoc_amd.types import OPT_CONDENSE_SPLIT
    default = ctx.get_option(OPT_CONDENSE_SPLIT)
    ctx.set_condense_split(not default)
    for _ in range(reps):
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_CDD, cdd)
        ctx.upload(BUF_CON, con)
        ctx.condense()
    ctx.set_condense_split(default)
    assert (ctx.status() == 0).all()
    # ... and the role-split one-kernel condensation (RTOC_OPT_CONDENSE_REGISTER = 0) where the register-chained kernel is the default
    ctx.set_condense_register(False)
    for _ in range(reps):
        ctx.upload(BUF_KKT, kkt)
        ctx.upload(BUF_CDD, cdd)
        ctx.upload(BUF_CON, con)
        ctx.condense()
    ctx.set_condense_register(True)
    assert (ctx.status() == 0).all()
    del kkt, cdd, con
    # ---- rigid-body linearisation of the same batch ----
    from robotoc_amd import robot_model as rm
    from robotoc_amd.types import BUF_SOL
    model = rm.load_named("anymal")
    ctx.set_robot_model(model)
    masks, flip = [], False
    for g in grids:
        masks.append(0b1111 if g.dimf == 12 else 0 if g.dimf == 0 else (0b0110 if flip else 0b1001))
        flip = flip != (g.dimf == 6)
    ctx.set_contact_schedule(np.array(masks, dtype=np.uint32), np.zeros((len(grids), 4, 3)))
    rng = np.random.default_rng(0)
    o = L.sol.off
    sol = np.zeros((uniq, len(grids), L.sol.stride))
    for b in range(uniq):
        for i in range(len(grids)):
            q, v, a = rm.random_configuration(model, rng, 0.8)
            sol[b, i, o[0]:o[0] + 19], sol[b, i, o[1]:o[1] + 18], sol[b, i, o[2]:o[2] + 18] = q, v, a
            sol[b, i, o[3]:o[3] + 12] = rng.uniform(-5, 5, 12)
            sol[b, i, o[4]:o[4] + 12] = rng.uniform(-20, 20, 12)
            sol[b, i, o[7]:o[7] + 18] = rng.uniform(-1, 1, 18)
            sol[b, i, o[8]:o[8] + 12] = rng.uniform(-1, 1, 12)
    ctx.upload(BUF_SOL, tile(sol, batch))
    for _ in range(reps):
        ctx.linearize_contact_dynamics(True)
    ctx.sync()
    ctx.close()
    # ---- ANYmal jump with switching-time optimisation, 4096 instances: the STO instantiation of the register-resident kernel ----
    dims, grids, _ = pr.config_anymal_jump_sto()
    batch = 4096
    ctx = capi.Context(dims, len(grids), batch, 0)
    ctx.set_grid(grids)
    ctx.upload(BUF_KKT, tile(pr.make_kkt_batch_unique(ctx.L, grids, uniq, seed=7), batch))
    ctx.upload(BUF_DX0, tile(pr.make_dx0_unique(ctx.L, uniq, seed=7), batch))
    for _ in range(reps):
        ctx.riccati_backward()
        ctx.riccati_forward()
    assert (ctx.status() == 0).all()
    ctx.close()
    # ---- iCub nv=32 / nv=35, 1024 instances: backward + forward, condense + expand (nv = 32: the register-wide backward kernel) ----
    for nv in (32, 35):
        dims, grids, _ = pr.config_icub_jump(nv=nv)
        dims = icub_dims(dims.nv, nc_max=(6 * dims.nu + 34 + 7) & ~7)   # room for the joint-limit and wrench-cone rows (as bench.py)
        batch = 1024
        ctx = capi.Context(dims, len(grids), batch, 0)
        L = ctx.L
        ctx.set_grid(grids)
        ctx.upload(BUF_KKT, tile(pr.make_kkt_batch_unique(L, grids, 16, seed=7), batch))
        ctx.upload(BUF_DX0, tile(pr.make_dx0_unique(L, 16, seed=7), batch))
        for _ in range(reps):
            ctx.riccati_backward()
            ctx.riccati_forward()
        assert (ctx.status() == 0).all()
        # the SQP iteration bench.py times for iCub: joint-limit rows and the 2 x 17 wrench-cone rows with both soles down
        ctx.set_constraint_rows(joint_limit_rows(dims))
        ctx.set_wrench_cones(2)
        cones = [capi.wrench_cone_matrix(0.1, 0.05, 0.6), capi.wrench_cone_matrix(0.09, 0.055, 0.7)]
        ctx.upload(BUF_CONE, pr.make_wrench_cone_batch(L, grids, batch, 2, cones))
        kkt, cdd = pr.make_precondense_batch_unique(L, grids, 16, seed=7)
        kkt, cdd = tile(kkt, batch), tile(cdd, batch)
        con = tile(pr.make_constraint_batch_unique(L, grids, 16, seed=7), batch)
        for _ in range(reps):
            ctx.upload(BUF_KKT, kkt)
            ctx.upload(BUF_CDD, cdd)
            ctx.upload(BUF_CON, con)
            ctx.condense()
            ctx.riccati_backward()
            ctx.riccati_forward()
            ctx.expand(0.995)
        print("iCub nv=%d: %d instances flagged (no data-dependent branches: the counters do not depend on it)" % (nv, int((ctx.status() != 0).sum())))
        ctx.close()
    print("pmc_driver done")


if __name__ == "__main__":
    main()
