#!/usr/bin/env python3
"""bench.py -- Riccati sweeps/s on the ANYmal trot workload (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: the backward Riccati recursion
followed by the forward recursion for PER_GPU_BATCH independent ANYmal trot OCP
instances (nv=18, 4 point contacts, N=40 -> 47 grids with 2 lifts + 2 impacts),
inputs resident in HBM.  One process per GPU; instances are sharded across ranks
with no data-path collective (weak scaling: fixed work per GPU); after the timed
region the step directions are all-gathered over RCCL (the one real exchange step,
SURVEY 8e) to validate the multi-GPU path.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PER_GPU_BATCH = 4096
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes(L, grids, batch, which):
    """algorithmic HBM bytes of one launch (dense-as-handed-over fields,
    P+ resident on chip), summed over the stages of the grid actually launched."""
    from robotoc_amd.types import GRID_IMPACT
    d = L.dims
    nv, nu, nx = d.nv, d.nu, 2 * d.nv
    total = 0
    N = len(grids) - 1
    for i, g in enumerate(grids):
        if i == N:
            if which == "backward":
                total += 2 * (nx * nx + nx)  # read Qxx,lx; write P,s
            else:
                total += nx * nx + nx + nx + nx  # read P,s,dx ; write dlmdgmm
            continue
        ns = g.dims
        if which == "backward":
            if g.type == GRID_IMPACT:
                rd = 2 * nx * nx + 2 * nx
                wr = nx * nx + nx
            else:
                rd = 2 * nx * nx + nx * nu + nu * nu + nv * nu + 2 * nx + nu
                wr = nx * nx + nx + nu * nx + nu
                if ns:
                    rd += ns * nx + ns * nu + ns
                    wr += ns * nx + ns
            total += rd + wr
        else:
            if g.type == GRID_IMPACT:
                rd = 2 * nx * nx + 2 * nx
                wr = 2 * nx
            else:
                rd = 2 * nx * nx + nu * nx + nv * nu + 2 * nx + nu
                wr = 2 * nx + nu
                if ns:
                    rd += ns * nx + ns
                    wr += ns
            total += rd + wr
    return total * 8 * batch


def condense_bytes(L, grids, batch):
    """Algorithmic HBM bytes of one rtoc_condense launch: per non-terminal grid point the
    ContactDynamicsData inputs and the un-condensed Hessian / gradient blocks are read once, the condensed
    blocks, the new dynamics rows and the data the expansion needs (MJtJinv, MJtJinv_dIDCdqv, Qafqv,
    Qafu) are written once; max-size backing blocks are moved whole, like the reference stores them."""
    from robotoc_amd.types import GRID_IMPACT, GRID_TERMINAL
    d = L.dims
    nv, nu, nx, nf, npas = d.nv, d.nu, 2 * d.nv, d.nf_max, d.np
    nvf = nv + nf
    total = 0
    for g in grids:
        if g.type == GRID_TERMINAL:
            continue
        imp = g.type == GRID_IMPACT
        rd = nv * nv + nvf * nx + nvf + nv + nf * nf + nv * nf + 2 * nvf + nx * nx + 2 * nx + nv
        wr = nx * nx + nv * nx + nx + nv + nvf * nvf + 2 * nvf * nx + 2 * nvf
        if not imp:
            rd += nf * nv + nx * nu + nu * nu + 2 * nx + 2 * nu + 2 * nvf
            wr += nx * nu + nu * nu + nv * nu + 2 * nx + 2 * nu + nvf * nv + nx * npas + npas * nu + nvf
            if g.dims:
                rd += g.dims * (nv + nx + 2)
                wr += g.dims * (nx + nu + 2)
        total += rd + wr
    return total * 8 * batch


def pmc_traffic(waves, batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (profiles/r01_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, separate --pmc runs).
    PMC counters cannot be collected from inside the timed process; null if the committed pass
    does not match the configuration being run."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if waves not in (0, 8) or batch != PER_GPU_BATCH or not os.path.exists(path):
        return None
    t = json.load(open(path))
    for k, v in t.items():
        if "backward_rs" in k:
            return v["hbm_bytes"]
    return None


def cpu_baseline(L, grids, dx0_small, kkt_small, budget_s=12.0):
    """Oracle (CPU port of the reference algorithm) timed on this box's host cores, OpenMP over
    instances.  Bounded sample: repeat a small batch until ~budget_s of CPU work."""
    from oracle import oracle as orc
    from robotoc_amd.types import Records
    nthreads = os.cpu_count() or 1
    # at least two instances per hardware thread so that every core has work
    reps_b = max(1, (2 * nthreads + kkt_small.shape[0] - 1) // kkt_small.shape[0])
    kkt_small = np.ascontiguousarray(np.tile(kkt_small, (reps_b, 1, 1)))
    dx0_small = np.ascontiguousarray(np.tile(dx0_small, (reps_b, 1)))
    B = kkt_small.shape[0]
    ric = Records(L, "ric").zeros(B, len(grids))
    d = Records(L, "dir").zeros(B, len(grids))
    work = kkt_small.copy()
    orc.riccati_sweep_batch(L, grids, work, ric, d, dx0=dx0_small)  # warm-up (page faults)
    t0 = time.perf_counter()
    reps = 0
    while True:
        work[...] = kkt_small  # the oracle mutates the KKT blocks in place like the reference
        orc.riccati_sweep_batch(L, grids, work, ric, d, dx0=dx0_small)
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    # the reference's Riccati recursion itself is single-threaded (riccati_recursion.cpp): one instance through
    # the single-instance entry points (no OpenMP region: forking a 256-thread team costs more than the sweep)
    one_k, one_d = np.ascontiguousarray(kkt_small[0]), np.ascontiguousarray(dx0_small[0])
    ric1, d1, w1 = Records(L, "ric").zeros(len(grids)), Records(L, "dir").zeros(len(grids)), one_k.copy()
    Dv = Records(L, "dir")
    t1 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t1 < 2.0:
        w1[...] = one_k
        Dv.f(d1[0], "dx")[...] = one_d
        orc.riccati_backward(L, grids, w1, ric1)
        orc.riccati_forward(L, grids, w1, ric1, d1)
        n1 += 1
    single = n1 / (time.perf_counter() - t1)
    return dict(value=B * reps / dt, unit="sweeps/s", cores=nthreads, kind="port",
                single_thread_sweeps_per_sec=single, single_thread_sweep_ms=1e3 / single,
                sample="%d instances x %d repeats of the same ANYmal trot sweep, OpenMP over "
                       "instances (%d threads), includes the memcpy that restores the in-place "
                       "mutated KKT blocks" % (B, reps, nthreads))


def sqp_single_instance(dims, grids, device):
    """SQP hot path of ONE OCP (the regime of a robotoc::OCPSolver call): per-phase HIP-event times with the
    serial recursions and with both recursions as horizon scans (RTOC_OPT_BACKWARD_SCAN)."""
    from robotoc_amd import capi, problems as pr
    from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_DX0, BUF_KKT, joint_limit_rows
    out = {}
    for mode in ("serial", "scan"):
        c1 = capi.Context(dims, len(grids), 1, device)
        L1 = c1.L
        c1.set_grid(grids)
        c1.set_constraint_rows(joint_limit_rows(dims))
        c1.set_friction_cones(4, 3)
        c1.set_backward_scan(mode == "scan")
        kkt, cdd = pr.make_precondense_batch(L1, grids, 1)
        con = pr.make_constraint_batch(L1, grids, 1)
        c1.upload(BUF_CONE, pr.make_cone_batch(L1, grids, 1, 4))
        c1.upload(BUF_DX0, pr.make_dx0(L1, 1))
        ph = {"condense": 2, "backward": 0, "forward": 1, "expand": 3, "update": 5}
        acc = {k: 0.0 for k in ph}
        nrep = 3
        for rep in range(nrep + 1):
            c1.upload(BUF_KKT, kkt)
            c1.upload(BUF_CDD, cdd)
            c1.upload(BUF_CON, con)
            for name in ("condense", "backward", "forward", "expand", "update"):
                ms = c1.time_phase(ph[name], 1)
                if rep > 0:
                    acc[name] += ms / nrep
        ok = int((c1.status() != 0).sum()) == 0
        c1.close()
        out[mode] = {"ms": acc, "total_ms": sum(acc.values()), "iters_per_sec": 1e3 / sum(acc.values()), "status_ok": ok}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="instances per GPU")
    ap.add_argument("--waves", type=int, default=0, help="backward-kernel waves per instance (0=default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sqp", action="store_true", help="skip the SQP-iteration phase timing")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json configurations")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("RTOC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0  # functional test of the multi-process path on a single-GPU box
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # "nccl" is RCCL on ROCm; RTOC_BENCH_BACKEND=gloo only for the single-GPU functional test of
        # this multi-process path (RCCL refuses two ranks on one device)
        dist.init_process_group(os.environ.get("RTOC_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    from robotoc_amd import capi, problems as pr
    from robotoc_amd.types import BUF_DIR, BUF_DX0, BUF_KKT

    dims, grids, _ = pr.config_anymal_trot()
    batch = args.batch
    ctx = capi.Context(dims, len(grids), batch, local_rank)
    L = ctx.L
    ctx.set_grid(grids)
    if args.waves:
        ctx.set_backward_waves(args.waves)
    # per-rank shard of the instance range: seeds are offset by rank*batch
    uniq = 16
    kkt_small = pr.make_kkt_batch(L, grids, uniq, first_instance=rank * batch)
    dx0_small = pr.make_dx0(L, uniq, first_instance=rank * batch)
    reps = (batch + uniq - 1) // uniq
    kkt = np.ascontiguousarray(np.tile(kkt_small, (reps, 1, 1))[:batch])
    dx0 = np.ascontiguousarray(np.tile(dx0_small, (reps, 1))[:batch])
    # the direction buffer lives in a torch tensor so that torch.distributed (RCCL) can gather it
    dir_t = torch.zeros(ctx.buffer_count(BUF_DIR), dtype=torch.float64, device="cuda")
    ctx.bind(BUF_DIR, dir_t.data_ptr())
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.upload(BUF_KKT, kkt)
    ctx.upload(BUF_DX0, dx0)

    def step():
        ctx.riccati_backward()
        ctx.riccati_forward()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    bad = int((ctx.status() != 0).sum())

    # attainable HBM bandwidth on this box: device-to-device copy of 1 GiB (read + write), same run
    src = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst

    # dominant kernel (backward) and forward timed with HIP events on the launch stream
    ms_b = ctx.time_phase(0, max(3, args.steps // 2))
    ms_f = ctx.time_phase(1, max(3, args.steps // 2))
    nsl = len(grids) * L.dir.stride
    gathered_ok = None
    if world > 1:
        from robotoc_amd.sharding import gather_directions
        full = gather_directions(dir_t[:batch * nsl].view(batch, len(grids), L.dir.stride),
                                 world * batch, world, rank)
        torch.cuda.synchronize()
        gathered_ok = bool(full.shape[0] == world * batch and torch.isfinite(full).all().item())

    # ---- SQP-iteration hot path (condense -> backward -> forward -> expand) on pre-condensation
    #      stage data; each phase timed with HIP events on the launch stream, the (untimed)
    #      device-to-device restore of the in-place-mutated records happens between repetitions ----
    sqp = None
    if not args.no_sqp:
        from robotoc_amd.types import BUF_CDD
        kkt_pre_s, cdd_pre_s = pr.make_precondense_batch(L, grids, 4, first_instance=rank * batch)
        rp = (batch + 3) // 4
        kkt0 = torch.from_numpy(np.ascontiguousarray(np.tile(kkt_pre_s, (rp, 1, 1))[:batch])).cuda()
        cdd0 = torch.from_numpy(np.ascontiguousarray(np.tile(cdd_pre_s, (rp, 1, 1))[:batch])).cuda()
        kkt_w = torch.empty(ctx.buffer_count(BUF_KKT), dtype=torch.float64, device="cuda")
        cdd_w = torch.empty(ctx.buffer_count(BUF_CDD), dtype=torch.float64, device="cuda")
        ctx.bind(BUF_KKT, kkt_w.data_ptr())
        ctx.bind(BUF_CDD, cdd_w.data_ptr())
        # PDIPM joint-limit rows (6 x nu box rows, examples/anymal/trot.cpp:134-146)
        from robotoc_amd.types import BUF_CON, joint_limit_rows
        ctx.set_constraint_rows(joint_limit_rows(dims))
        con_s = pr.make_constraint_batch(L, grids, 4, first_instance=rank * batch)
        con0 = torch.from_numpy(np.ascontiguousarray(np.tile(con_s, (rp, 1, 1))[:batch])).cuda()
        con_w = torch.empty(ctx.buffer_count(BUF_CON), dtype=torch.float64, device="cuda")
        ctx.bind(BUF_CON, con_w.data_ptr())
        # friction cones of the (up to 4) active point contacts: 5 PDIPM rows each, dense Jacobians
        from robotoc_amd.types import BUF_CONE
        ctx.set_friction_cones(4, 3)
        cone_s = pr.make_cone_batch(L, grids, 4, 4, first_instance=rank * batch)
        ctx.upload(BUF_CONE, np.ascontiguousarray(np.tile(cone_s, (rp, 1, 1))[:batch]))
        ph = {"condense": 2, "backward": 0, "forward": 1, "expand": 3, "update": 5}
        acc = {k: 0.0 for k in ph}
        nrep = 3
        for rep in range(nrep + 1):
            kkt_w[:kkt0.numel()].copy_(kkt0.view(-1))
            cdd_w[:cdd0.numel()].copy_(cdd0.view(-1))
            con_w[:con0.numel()].copy_(con0.view(-1))
            torch.cuda.synchronize()
            for name in ("condense", "backward", "forward", "expand", "update"):
                ms = ctx.time_phase(ph[name], 1)
                if rep > 0:
                    acc[name] += ms / nrep
        bad_sqp = int((ctx.status() != 0).sum())
        tot = sum(acc.values())
        cb = condense_bytes(L, grids, batch)
        sqp = {"ms": acc, "total_ms": tot, "iters_per_sec_per_gpu": batch / tot * 1e3,
               "single_instance": sqp_single_instance(dims, grids, local_rank) if rank == 0 else None,
               "condense_algorithmic_bytes": cb, "condense_GBs_algorithmic": cb / (acc["condense"] * 1e-3) / 1e9,
               "status_nonzero_instances": bad_sqp,
               "scope": "hot path downstream of the Pinocchio linearisation: PDIPM condensation of the "
                        "joint-limit and friction-cone rows + computeMJtJinv + condenseContact/ImpactDynamics + Riccati "
                        "backward/forward + expandContactDynamics primal/dual + PDIPM expansion, "
                        "fraction-to-boundary step sizes and slack/dual update; linearisation, cost "
                        "and the manifold update of q are CPU-side and excluded"}

    # ---- the other BASELINE.json configurations (parity-test cases; reported, not the headline):
    #      batch throughput and the single-instance sweep latency a robotoc OCPSolver call would see ----
    others = None
    if rank == 0 and not args.no_configs:
        others = {}
        cfgs = [("anymal_trot_N40", pr.config_anymal_trot, 0), ("anymal_jump_sto_N40", pr.config_anymal_jump_sto, 4096),
                ("icub_nv32_jump_N30", lambda: pr.config_icub_jump(nv=32), 1024),
                ("icub_nv35_jump_N30", lambda: pr.config_icub_jump(nv=35), 1024), ("iiwa14_unconstr_N20", pr.config_iiwa14, 4096)]
        for name, fn, nb in cfgs:
            d2, g2, info = fn()
            entry = {"stages": len(g2)}
            for label, b2 in (("single_instance", 1), ("batch", nb)):
                if b2 == 0:
                    continue
                c2 = capi.Context(d2, len(g2), b2, local_rank)
                L2 = c2.L
                c2.set_grid(g2)
                if name.startswith("iiwa"):
                    from robotoc_amd.types import Records
                    k1 = Records(L2, "kkt").zeros(1, len(g2))
                    pr.fill_unconstr_instance(L2, len(g2), k1[0], np.random.default_rng(1))
                    c2.upload(BUF_KKT, np.ascontiguousarray(np.tile(k1, (b2, 1, 1))))
                    c2.unconstr_backward(info["dt"])  # materialises the structured A, B once
                else:
                    c2.upload(BUF_KKT, pr.make_kkt_batch_tiled(L2, g2, b2, unique=min(b2, 4)))
                c2.upload(BUF_DX0, np.ascontiguousarray(np.tile(pr.make_dx0(L2, 1), (b2, 1))))
                c2.time_phase(4, 1)
                mb, mf = c2.time_phase(0, 3), c2.time_phase(1, 3)
                ok = int((c2.status() != 0).sum()) == 0
                if b2 == 1:
                    entry["single_instance_sweep_ms"] = mb + mf
                    entry["single_instance_backward_ms"] = mb
                    if not any(g.sto or g.sto_next for g in g2):
                        # RTOC_OPT_BACKWARD_SCAN: both recursions as scans over the horizon (latency path)
                        c2.set_backward_scan(True)
                        c2.time_phase(4, 2)
                        ms, msf = c2.time_phase(0, 5), c2.time_phase(1, 5)
                        c2.set_backward_scan(False)
                        entry["single_instance_backward_scan_ms"] = ms
                        entry["single_instance_forward_scan_ms"] = msf
                        entry["single_instance_sweep_scan_ms"] = ms + msf
                        ok = ok and int((c2.status() != 0).sum()) == 0
                else:
                    entry.update({"batch": b2, "backward_ms": mb, "forward_ms": mf,
                                  "sweeps_per_sec": b2 / (mb + mf) * 1e3,
                                  "backward_GBs_algorithmic": algorithmic_bytes(L2, g2, b2, "backward") / mb / 1e6})
                entry["status_ok"] = entry.get("status_ok", True) and ok
                c2.close()
            others[name] = entry

    if rank == 0:
        total_sweeps = world * batch * args.steps
        value = total_sweeps / dt
        bytes_b = algorithmic_bytes(L, grids, batch, "backward")
        bytes_f = algorithmic_bytes(L, grids, batch, "forward")
        ach = bytes_b / (ms_b * 1e-3) / 1e9
        res = {
            "metric": "riccati_sweeps_per_sec",
            "value": value,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "anymal_trot_N40 (nv=18,nu=12, 47 grids: 2 lifts + 2 impacts, "
                                   "switching constraints ns=6), %d OCP instances per GPU, "
                                   "backward+forward Riccati sweep" % batch,
                       "per_gpu_batch": batch, "stages": len(grids), "parallelism": "instances sharded, dp%d" % world,
                       "backward_waves": args.waves},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(args.waves, batch),
                         "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": ach / copy_gbs,
                         "kernel": "riccati_backward_rs4_kernel" if args.waves in (0, 8) else "riccati_backward_kernel", "kernel_ms": ms_b,
                         "algorithmic_bytes_per_launch": bytes_b,
                         "forward_kernel_ms": ms_f,
                         "forward_achieved": bytes_f / (ms_f * 1e-3) / 1e9,
                         "forward_algorithmic_bytes_per_launch": bytes_f},
            "status_nonzero_instances": bad,
        }
        if sqp is not None:
            res["sqp_iteration"] = sqp
        if others is not None:
            res["other_configs"] = others
        if gathered_ok is not None:
            res["rccl_gather_ok"] = gathered_ok
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(L, grids, dx0_small, kkt_small)
        print(json.dumps(res))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
