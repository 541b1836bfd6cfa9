"""Run-to-run determinism of every kernel with an inter-wave hand-off (profiles/HISTORY.md 3.1 (e)), on the GPU.

A hand-off race shows as a wrong record in a handful of instance-sweeps out of 100k: one sweep compared with the oracle passes
nine times out of ten with it.  These tests repeat the same launch on one context and compare EVERY downloaded record --
the Riccati factorisation (RTOC_BUF_RIC) as well as the directions (RTOC_BUF_DIR) -- with the first run's, bit for bit.

protocol under test                                   case
  register-resident kernel (LDS-DMA, deferred P stores)  trot-register         (riccati_backward_rv.hpp, the default for this shape)
  rs4 role-split, structured Fxx (flag words, rs_sync)  trot-structured       (riccati_backward_rs.hpp, SA = true)
  rs4 role-split, dense-Fxx fallback                    trot-dense            (SA = false: RTOC_OPT_FXX_STRUCTURE = 1)
  rs4 + STO block + phase transition                    jump_sto              (riccati_sto_block.inc, riccati_pt_block.inc)
  register-wide kernels (default dispatch at 1024)      icub32, icub35        (riccati_backward_rw.hpp; _rw2.hpp: two waves per instance,
                                                                               hand-over of Y / H^T / F through LDS behind s_barrier)
  horizon scan (element / combine kernels + vector pass) trot-scan, jump_sto-scan (riccati_scan_core.hpp, riccati_scan_sto.hpp)
  one Newton iteration as one launch sequence           newton               (condense, sweep, expand, update from restored records)
"""
import numpy as np
import pytest

from robotoc_amd import problems as pr
from robotoc_amd.types import (BUF_CDD, BUF_CON, BUF_CONE, BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC, BUF_SOL, BUF_STEP,
                               joint_limit_rows)

CASES = {
    # name: (configuration, batch, repetitions, context set-up)
    "trot-register": (pr.config_anymal_trot, 4096, 30, lambda c: c.set_backward_register(True)),
    "trot-structured": (pr.config_anymal_trot, 4096, 30, lambda c: (c.set_backward_register(False), c.set_fxx_structure(2))),
    "trot-dense": (pr.config_anymal_trot, 4096, 30, lambda c: (c.set_backward_register(False), c.set_fxx_structure(1))),
    "jump_sto": (pr.config_anymal_jump_sto, 4096, 30, None),
    "icub32": (lambda: pr.config_icub_jump(nv=32), 1024, 20, None),
    "icub35": (lambda: pr.config_icub_jump(nv=35), 1024, 20, None),
    "trot-scan": (pr.config_anymal_trot, 16, 60, lambda c: c.set_backward_scan(True)),
    "jump_sto-scan": (pr.config_anymal_jump_sto, 16, 60, lambda c: c.set_backward_scan(True)),
}


def _dev_records(torch, L, which, batch, n):
    return torch.zeros((batch, n, getattr(L, which).stride), dtype=torch.float64, device="cuda:0")


def _first_difference(torch, name, a, b):
    """None when the two device tensors are bit-identical, else where they differ (compared as 64-bit words on the device:
    the Riccati records of a headline batch are 3.9 GB, too much to download 30 times)."""
    ne = a.view(torch.int64) != b.view(torch.int64)
    if not bool(ne.any()):
        return None
    idx = ne.nonzero()
    return "%s: %d words differ, instances %s, stages %d..%d" % (
        name, int(ne.sum()), sorted(set(idx[:, 0].tolist()))[:8], int(idx[:, 1].min()), int(idx[:, 1].max()))


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_sweep_repeats_bit_for_bit(case):
    import torch
    from robotoc_amd import capi
    cfg, batch, reps, setup = CASES[case]
    dims, grids, _ = cfg()
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L, n = ctx.L, len(grids)
        ctx.set_grid(grids)
        if setup:
            setup(ctx)
        kkt = pr.make_kkt_batch_unique(L, grids, batch, backend="torch", device="cuda:0", out=_dev_records(torch, L, "kkt", batch, n))
        dx0 = pr.make_dx0_unique(L, batch, backend="torch", device="cuda:0").contiguous()
        ric, d = _dev_records(torch, L, "ric", batch, n), _dev_records(torch, L, "dir", batch, n)
        for buf, t in ((BUF_KKT, kkt), (BUF_DX0, dx0), (BUF_RIC, ric), (BUF_DIR, d)):
            ctx.bind(buf, t.data_ptr())
        torch.cuda.synchronize()
        first = None
        for run in range(reps):
            ctx.riccati_backward()
            ctx.riccati_forward()
            ctx.sync()
            if first is None:
                first = (ric.clone(), d.clone())
                assert bool(first[0].abs().sum() > 0) and bool(first[1].abs().sum() > 0)
                ric.zero_()
                d.zero_()
                torch.cuda.synchronize()
                continue
            for name, a, b in (("ric", ric, first[0]), ("dir", d, first[1])):
                msg = _first_difference(torch, name, a, b)
                assert msg is None, "%s, run %d: %s" % (case, run, msg)
        assert (ctx.status() == 0).all()
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("split", [1, 0])
def test_newton_iteration_repeats_bit_for_bit(split):
    """30 x rtoc_newton_iteration (condensation incl. joint-limit and friction-cone rows, sweep, expansion, step sizes, update,
    integration) from identical restored records: every buffer the iteration writes is compared with the first run's.
    split: RTOC_OPT_CONDENSE_SPLIT (two-kernel / fused condensation)."""
    import torch
    from robotoc_amd import capi
    MC, CD, batch, tau = 4, 3, 1024, 0.995
    dims, grids, _ = pr.config_anymal_trot()
    ctx = capi.Context(dims, len(grids), batch, 0)
    try:
        L, n, dev = ctx.L, len(grids), "cuda:0"
        ctx.set_grid(grids)
        ctx.set_condense_split(split)
        ctx.set_constraint_rows(joint_limit_rows(dims))
        ctx.set_friction_cones(MC, CD)
        rec = lambda which: _dev_records(torch, L, which, batch, n)
        kkt0, cdd0 = pr.make_precondense_batch_unique(L, grids, batch, backend="torch", device=dev, out=(rec("kkt"), rec("cdd")))
        con0 = pr.make_constraint_batch_unique(L, grids, batch, backend="torch", device=dev, out=rec("con"))
        cone = pr.make_cone_batch_unique(L, grids, batch, MC, backend="torch", device=dev).contiguous()
        dx0 = pr.make_dx0_unique(L, batch, backend="torch", device=dev).contiguous()
        gen = torch.Generator(device=dev)
        gen.manual_seed(5)
        sol0 = 2.0 * torch.rand(batch, n, L.sol.stride, dtype=torch.float64, device=dev, generator=gen) - 1.0
        o = L.sol.off[0]
        sol0[:, :, o + 3:o + 7] /= sol0[:, :, o + 3:o + 7].norm(dim=-1, keepdim=True)  # q on the manifold
        work = dict(kkt=torch.empty_like(kkt0), cdd=torch.empty_like(cdd0), con=torch.empty_like(con0), sol=torch.empty_like(sol0),
                    ric=rec("ric"), dir=rec("dir"))
        for buf, t in ((BUF_KKT, work["kkt"]), (BUF_CDD, work["cdd"]), (BUF_CON, work["con"]), (BUF_SOL, work["sol"]),
                       (BUF_RIC, work["ric"]), (BUF_DIR, work["dir"]), (BUF_CONE, cone), (BUF_DX0, dx0)):
            ctx.bind(buf, t.data_ptr())
        first = None
        for run in range(30):
            for k, src in (("kkt", kkt0), ("cdd", cdd0), ("con", con0), ("sol", sol0)):
                work[k].copy_(src)
            work["ric"].zero_()
            work["dir"].zero_()
            torch.cuda.synchronize()
            ctx.newton_iteration(0.0, tau)  # tolerance 0: nobody is converged, every instance steps
            ctx.sync()
            steps = torch.from_numpy(ctx.download(BUF_STEP, (batch, 2)).reshape(batch, 1, 2).copy()).to(dev)
            got = {k: v.clone() for k, v in work.items()}
            got["steps"] = steps
            if first is None:
                first = got
                assert bool(got["dir"].abs().sum() > 0)
                continue
            for name in sorted(got):
                msg = _first_difference(torch, name, got[name], first[name])
                assert msg is None, "run %d: %s" % (run, msg)
        assert (ctx.status() == 0).all()
    finally:
        ctx.close()
