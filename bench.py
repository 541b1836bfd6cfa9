#!/usr/bin/env python3
"""bench.py -- Riccati sweeps/s on the ANYmal trot workload (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: the backward Riccati recursion
followed by the forward recursion for PER_GPU_BATCH independent, DISTINCT ANYmal trot OCP
instances (nv=18, 4 point contacts, N=40 -> 47 grids with 2 lifts + 2 impacts; randomised
stage data and initial state directions, generated in HBM), inputs resident in HBM.
One process per GPU; instances are sharded across ranks with no data-path collective
(weak scaling: fixed work per GPU); after the timed region the step directions are
all-gathered over RCCL (the one real exchange step) to validate the multi-GPU path.

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
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X spec sheet, dense fp64 matrix (= the fp64 vector rate)
PROFILE_ROUND = "r06"


def kernel_source_hash():
    """sha256 over the kernel sources the library is built from (robotoc_amd/csrc/*.hpp|*.inc|*.hip, include/*.h): the
    committed counter passes (profiles/<round>_traffic.json, written by tools/pmc_driver.py) carry the hash of the sources
    they profiled, and a pass taken from other sources is refused (traffic = null) instead of silently going stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "robotoc_amd", "csrc", "*.hpp")) + glob.glob(os.path.join(ROOT, "robotoc_amd", "csrc", "*.inc"))
                   + glob.glob(os.path.join(ROOT, "robotoc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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


def backward_flops(L, grids, batch):
    """SURVEY 8(d) flop count of one backward launch (the reference's operation count, dense Fxx):
    4nx^3 + 2nx^2 nu + 4nu^2 nx + 4nx nv nu + 2nu^2 nv + nu^3/3 + 4nx^2 + 2nx nu per control stage,
    4nx^3 + 4nx^2 per impact stage."""
    from robotoc_amd.types import GRID_IMPACT, GRID_TERMINAL
    d = L.dims
    nv, nu, nx = d.nv, d.nu, 2 * d.nv
    total = 0.0
    for g in grids:
        if g.type == GRID_TERMINAL:
            continue
        if g.type == GRID_IMPACT:
            total += 4 * nx ** 3 + 4 * nx ** 2
        else:
            total += (4 * nx ** 3 + 2 * nx * nx * nu + 4 * nu * nu * nx + 4 * nx * nv * nu + 2 * nu * nu * nv
                      + nu ** 3 / 3.0 + 4 * nx * nx + 2 * nx * nu)
    return total * batch


def condense_bytes(L, grids, batch):
    """Algorithmic HBM bytes of one rtoc_condense launch: per non-terminal grid point the
    ContactDynamicsData inputs and the un-condensed Hessian / gradient blocks are read once, the condensed
    blocks, the new dynamics rows and the data the expansion needs (MJtJinv, MJtJinv_dIDCdqv, MJtJinv_IDC, laf,
    haf) are written once; max-size backing blocks are moved whole, like the reference stores them.  Qafqv and
    Qafu_full -- scratch the reference keeps for expandContactDynamicsDual -- are not counted since round 2:
    rtoc_expand rebuilds their products from Qaa, Qff, Qqf (RTOC_OPT_CONDENSE_KEEP_QAF)."""
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
        wr = nx * nx + nv * nx + nx + nv + nvf * nvf + nvf * nx + 2 * nvf  # (Qafqv is not stored: RTOC_OPT_CONDENSE_KEEP_QAF = 0)
        if not imp:
            rd += nf * nv + nx * nu + nu * nu + 2 * nx + 2 * nu + 2 * nvf
            wr += nx * nu + nu * nu + nv * nu + 2 * nx + 2 * nu + nx * npas + npas * nu + nvf  # (nor Qafu_full)
            if g.dims:
                rd += g.dims * (nv + nx + 2)
                wr += g.dims * (nx + nu + 2)
        total += rd + wr
    return total * 8 * batch


def constraint_row_bytes(grids, batch, rows, cone_contacts, nv):
    """HBM bytes of the PDIPM rows' own data that one rtoc_condense launch moves: per active joint-limit row slack, dual, residual,
    cmpl in and cond out; per active friction cone (5 rows) the same plus dg_dq (5 x nv) and dg_df (5 x 3).  Reported beside the
    algorithmic bytes of the contact-dynamics condensation (condense_bytes), not inside them."""
    from robotoc_amd.types import GRID_IMPACT, GRID_TERMINAL
    total = 0
    for g in grids:
        if g.type == GRID_TERMINAL:
            continue
        if g.type != GRID_IMPACT:
            total += 5 * sum(1 for r in rows if g.time_stage >= r.level)
        nc = min(g.dimf // 3, cone_contacts)
        total += nc * (5 * 5 + 5 * nv + 15)
    return total * 8 * batch


def expand_bytes(L, grids, batch, rows, cone_contacts, wrench=False):
    """Algorithmic HBM bytes of one rtoc_expand launch: per non-terminal grid point what
    expandContactDynamicsPrimal/Dual need (contact_dynamics.cpp:167-202: MJtJinv, MJtJinv_dIDCdqv,
    MJtJinv_IDC, laf, haf, the passive blocks, Phia, and Qaa / Qff / Qqf from which Qafqv dx + Qafu du is rebuilt;
    dx, du, dgmm+, dxi) and write
    (daf, dbetamu, dnu_passive, laf), active views only, plus the PDIPM rows of
    Constraints::expandSlackAndDual (slack, dual, residual, cmpl in; dslack, ddual out; the cone rows
    also read their Jacobians)."""
    from robotoc_amd.types import GRID_IMPACT, GRID_TERMINAL
    d = L.dims
    nv, nu, nx, npas = d.nv, d.nu, 2 * d.nv, d.np
    total = 0
    for g in grids:
        if g.type == GRID_TERMINAL:
            continue
        imp = g.type == GRID_IMPACT
        nvf = nv + g.dimf
        rd = nvf * nx + nvf * nvf + 3 * nvf + nx + nv + nv + g.dimf * g.dimf + nv * g.dimf  # MJtJinv_dIDCdqv, MJtJinv, vectors, Qaa, Qff, Qqf
        wr = 3 * nvf
        if not imp:
            rd += nu + nx * npas + npas * nu + npas + g.dims * (nv + 1)
            wr += npas
            act = sum(1 for r in rows if g.time_stage >= r.level)  # stage mask, constraints_data.cpp:20-45
            rd += 4 * act
            wr += 2 * act
        if wrench:   # 17 rows per active surface contact: slack, dual, residual, cmpl and the 17 x 6 cone matrix in, dslack, ddual out
            nc = min(g.dimf // 6, cone_contacts)
            rd += nc * (4 * 17 + 17 * 6)
            wr += nc * 2 * 17
        else:
            nc = min(g.dimf // 3, cone_contacts)
            rd += nc * (4 * 5 + 5 * nv + 15)
            wr += nc * 2 * 5
        total += rd + wr
    return total * 8 * batch


def counter_pass_valid(table):
    """A committed counter pass belongs to this library if it carries the hash of the library's kernel sources -- or names
    that hash in its "_carried_over" list: an entry {"to": hash, "change": text} written BY HAND next to the commit that
    changed the sources, stating why the change cannot move the counted quantity (e.g. a flag wait moved inside one wave:
    no HBM access added or removed).  Returns (valid, note); the note goes into the bench line, so a carried-over figure is
    never presented as measured on these sources."""
    h = kernel_source_hash()
    if table.get("_kernel_source_hash") == h:
        return True, "kernel-source hash %s matches the library's sources" % h
    for c in table.get("_carried_over", []):
        if c.get("to") == h:
            return True, ("measured on kernel sources %s and CARRIED OVER to %s (%s)" % (table.get("_kernel_source_hash"), h, c.get("change")))
    return False, "kernel-source hash %s DOES NOT match the sources of this library (%s): refused as stale" % (table.get("_kernel_source_hash"), h)


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, separate --pmc runs, the guide's gfx950
    correction).  PMC counters cannot be collected from inside the timed process; null if no committed
    pass names the kernel."""
    path = os.path.join(ROOT, "profiles", "%s_traffic.json" % PROFILE_ROUND)
    if not os.path.exists(path):
        return None
    table = json.load(open(path))
    if not counter_pass_valid(table)[0]:
        return None   # counters of other kernel sources: stale (traffic_source says so)
    for k, v in table.items():
        if not k.startswith("_") and kernel_substr in k:
            return v["hbm_bytes"]
    return None


def rocprof_kernel_ms(kernel_substr):
    """{avg_ms, calls, back_to_back_avg_ms, ...} of a kernel over ALL launches of its largest geometry in the committed
    rocprofv3 --kernel-trace of this same command (profiles/<round>_traffic.json: _kernel_ms_rocprof, written by
    tools/summarize_profiles.py), or None: printed beside the event-timed kernel_ms of the line, which times the sweep loop only."""
    path = os.path.join(ROOT, "profiles", "%s_traffic.json" % PROFILE_ROUND)
    if not os.path.exists(path):
        return None
    table = json.load(open(path))
    if not counter_pass_valid(table)[0]:
        return None
    for k, v in table.get("_kernel_ms_rocprof", {}).items():
        if kernel_substr in k:
            return v
    return None


VALU_F64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (MI355X_MICROARCH.md: 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 x 2.4 GHz)


def linearize_flops():
    """fp64 VALU flops per grid point of the rigid-body linearisation, from the committed counter pass
    (profiles/<round>_linearize_flops.json, tools/gpu_pmc_lin_flops.sh): 64 x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-instructions
    of rbd_values_kernel + linearize_contact_dynamics_kernel.  (None, why) when no pass of THESE kernel sources is committed."""
    path = os.path.join(ROOT, "profiles", "%s_linearize_flops.json" % PROFILE_ROUND)
    if not os.path.exists(path):
        return None, "no committed fp64-instruction counter pass for this round"
    t = json.load(open(path))
    ok, note = counter_pass_valid(t)
    if not ok:
        return None, "profiles/%s_linearize_flops.json: %s" % (PROFILE_ROUND, note)
    return t["flops_per_grid_point"], ("profiles/%s_linearize_flops.json: rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 x 64 lanes, every "
                                       "lane slot counted, at %d grid points; %s" % (PROFILE_ROUND, t["grid_points"], note))


def traffic_state():
    path = os.path.join(ROOT, "profiles", "%s_traffic.json" % PROFILE_ROUND)
    if not os.path.exists(path):
        return "no committed counter pass for this round: traffic null"
    ok, note = counter_pass_valid(json.load(open(path)))
    return ("profiles/%s_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of the same kernels at the same "
            "sizes (tools/gpu_pmc2.sh -> tools/pmc_driver.py); %s%s" % (PROFILE_ROUND, note, "" if ok else ", traffic null"))


def cpu_baseline(L, grids, dims, budget_s=8.0):
    """Oracle (CPU port of the reference algorithm) timed on this box's host cores: Riccati sweeps and SQP
    hot-path iterations, one thread (the reference's Riccati recursion is single-threaded,
    riccati_recursion.cpp) and all threads (OpenMP over instances -- the reference's own parallelism is
    OpenMP over stages of evalKKT, direct_multiple_shooting.cpp:135).  Thread-private working records
    (oracle/rtoc_oracle_bench.c); bounded sample: distinct instances repeated until ~budget_s per leg."""
    from oracle import oracle as orc
    from robotoc_amd import problems as pr
    from robotoc_amd.types import joint_limit_rows
    hw_threads = os.cpu_count() or 1
    quota = hw_threads
    try:  # cgroup v2 CPU quota of the container: oversubscribing it gets the whole team throttled
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    try:
        quota = min(quota, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    nthreads = min(hw_threads, quota)
    B = max(4 * nthreads, 64)
    kkt = pr.make_kkt_batch_unique(L, grids, B, seed=99)
    dx0 = pr.make_dx0_unique(L, B, seed=99)
    out = {}

    def best_of(fn, n_items, nt, guess_s, legs=3):
        """Warm up (thread team, page faults, clocks), then `legs` measurements of ~budget_s/legs each;
        the fastest one is reported (the others are disturbed by whatever else the host is doing)."""
        fn(1, nt)
        w = fn(max(1, int(0.5 / max(guess_s * n_items / max(nt, 1), 1e-4))), nt)
        per_rep = w["seconds"] / max(w[[k for k in w if k in ("sweeps", "iterations")][0]] / n_items, 1)
        reps = max(1, int(budget_s / legs / max(per_rep, 1e-5)))
        runs = [fn(reps, nt) for _ in range(legs)]
        return min(runs, key=lambda r: r["seconds"]), reps

    # the restatement built -O3 -march=native ON THIS HOST (the portable x86-64-v3 object that travels with the
    # snapshot stays the parity suite's); falls back to the portable object, and says so, where gcc is missing
    native, build = True, None
    try:
        build = orc.native_lib()[1]
    except Exception as e:   # noqa: BLE001 -- no compiler on this host
        native, build = False, dict(flags="-O3 -march=x86-64-v3 -fopenmp (portable object; native build failed: %s)" % type(e).__name__)
    # ... and whichever of the two builds is FASTER on this host is the baseline (on the AVX-512 EPYC of the GPU boxes gcc's
    # -march=native code ran 0.89-0.91 k sweeps/s per thread against the x86-64-v3 object's 1.18 k: the native figure alone would
    # flatter the GPU); both single-thread rates are reported
    builds_1t = {}
    if native:
        for nat in (False, True):
            r_, _ = best_of(lambda reps, nt: orc.bench_sweep(L, grids, kkt[:16], dx0[:16], reps, nt, native=nat), 16, 1, 1.5e-3, legs=2)
            builds_1t["native" if nat else "x86-64-v3"] = r_["sweeps"] / r_["seconds"]
        if builds_1t["x86-64-v3"] > builds_1t["native"]:
            native = False
            build = dict(build, flags="-O3 -march=x86-64-v3 -fopenmp (faster on this host than -O3 -march=native: %.0f vs %.0f sweeps/s on one thread)"
                         % (builds_1t["x86-64-v3"], builds_1t["native"]))
        build = dict(build, single_thread_sweeps_per_sec_by_build=builds_1t)
    bench_sweep = lambda *a: orc.bench_sweep(*a, native=native)   # noqa: E731
    one, _ = best_of(lambda reps, nt: bench_sweep(L, grids, kkt[:16], dx0[:16], reps, nt), 16, 1, 1.5e-3, legs=2)
    # the quota is enforced per scheduling period: a team of up to 2x the quota can still come out ahead (SMT,
    # bursts) -- both are measured, the better one is reported with the thread count it used
    cands = [best_of(lambda reps, nt: bench_sweep(L, grids, kkt, dx0, reps, nt), B, nt_, 1.5e-3)
             for nt_ in sorted({nthreads, min(hw_threads, 2 * nthreads)})]
    allt, reps = max(cands, key=lambda c: c[0]["sweeps"] / c[0]["seconds"])
    nthreads = allt["threads"]
    out.update(value=allt["sweeps"] / allt["seconds"], unit="sweeps/s", cores=allt["threads"], kind="port",
               value_excluding_refill=allt["sweeps"] / max(allt["seconds"] - allt["refill_seconds"], 1e-9),
               single_thread_sweeps_per_sec=one["sweeps"] / one["seconds"],
               single_thread_sweep_ms=1e3 * one["seconds"] / one["sweeps"],
               single_thread_sweep_ms_excluding_refill=1e3 * (one["seconds"] - one["refill_seconds"]) / one["sweeps"],
               host_hardware_threads=hw_threads, cpu_quota=quota, build=build,
               sample_short="%d distinct ANYmal trot N40 instances x %d repeats on %d threads; 1 thread: %d sweeps" % (B, reps, allt["threads"], one["sweeps"]),
               sample="%d distinct ANYmal trot instances x %d repeats, OpenMP over instances (%d threads), every "
                      "thread refills a private copy of the in-place-mutated KKT records per sweep (that memcpy is "
                      "%.1f%% of the time; `value_excluding_refill` leaves it out); one thread: %d sweeps"
                      % (B, reps, allt["threads"], 100.0 * allt["refill_seconds"] / allt["seconds"], one["sweeps"]))
    # The reference's OWN RiccatiRecursion (src/riccati/riccati_recursion.cpp:32-131), timed the way OCPBenchmarker times
    # (include/robotoc/utils/ocp_benchmarker.hxx:14-32), one thread, on the same records -- for what it is: the reference sources
    # compiled -O2 with assertions behind the eager Eigen stand-in (oracle/ref_shim/mini_eigen.hpp; Eigen is not in the image),
    # so the expression templates Eigen would fuse and vectorise run as scalar loops with temporaries.  A floor for the
    # reference's speed, not its Eigen build.
    try:
        from oracle import ref
        if ref.available():
            ref.riccati_sweep_bench(L, grids, kkt[0], dx0[0], 1)
            w = ref.riccati_sweep_bench(L, grids, kkt[0], dx0[0], 8)
            reps_r = max(8, int(3.0 / max(w["seconds"] / 8, 1e-5)))
            runs = [ref.riccati_sweep_bench(L, grids, kkt[i % B], dx0[i % B], reps_r // 2) for i in range(2)]
            r = min(runs, key=lambda x: x["seconds"])
            out["reference_sources"] = dict(
                sweeps_per_sec=r["sweeps"] / r["seconds"], sweep_ms=1e3 * r["seconds"] / r["sweeps"], cores=1,
                sample="%d sweeps of one ANYmal trot N40 instance" % r["sweeps"],
                what="robotoc's own src/riccati (RiccatiRecursion::backward+forwardRiccatiRecursion) compiled in place, -O2, "
                     "asserts on, Eigen expressions evaluated eagerly by oracle/ref_shim/mini_eigen.hpp (Eigen absent): not an Eigen build")
    except Exception as e:   # noqa: BLE001 -- the checker library is optional here
        out["reference_sources"] = dict(error="%s: %s" % (type(e).__name__, e))
    # SQP hot-path iteration (both sides on identical pre-condensation inputs)
    Bs = max(nthreads, 32)
    kk, cc = pr.make_precondense_batch_unique(L, grids, Bs, seed=99)
    con = pr.make_constraint_batch_unique(L, grids, Bs, seed=99)
    cone = pr.make_cone_batch_unique(L, grids, Bs, 4, seed=99)
    rows = joint_limit_rows(dims)

    def sqp(n, reps, nt):
        return orc.bench_sqp(L, grids, kk[:n], cc[:n], con[:n], cone[:n], dx0[:n], rows, 4, 3, 0.995, reps, nt, native=native)
    s1, _ = best_of(lambda reps, nt: sqp(8, reps, nt), 8, 1, 6e-3, legs=2)
    sa, reps = best_of(lambda reps, nt: sqp(Bs, reps, nt), Bs, nthreads, 6e-3)
    out["sqp_iteration"] = dict(iters_per_sec=sa["iterations"] / sa["seconds"], threads=sa["threads"],
                                single_thread_iters_per_sec=s1["iterations"] / s1["seconds"],
                                single_thread_iter_ms=1e3 * s1["seconds"] / s1["iterations"],
                                sample="%d distinct instances x %d repeats (all threads), %d iterations (one thread); "
                                       "72 joint-limit + 20 friction-cone rows" % (Bs, reps, s1["iterations"]))
    return out


COMPACT_LIMIT = 4096   # bytes; the driver keeps an 8 KB tail of stdout and parses its last line


def _r(x, sig=6):
    """Round a float to `sig` significant digits (the compact line has no use for 17)."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    return x


def compact_line(res):
    """The ONE line the driver parses (<= COMPACT_LIMIT bytes, strict JSON): the contract keys, `roofline` and `cpu_baseline`
    of the headline workload, and one figure per other BASELINE configuration -- the convention of the reference's own
    benchmark printer, include/robotoc/utils/ocp_benchmarker.hxx:14-32 (one short figure per run).  Everything else
    (per-phase times, every other roofline block, the scopes and notes) is in the detail file."""
    def pick(d, keys):
        return {k: _r(d[k]) for k in keys if d is not None and k in d}
    out = pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                     "vs_baseline", "dtype", "data"))
    cfg = res.get("config", {})
    out["config"] = pick(cfg, ("workload", "per_gpu_batch", "distinct_instances_per_gpu", "stages", "parallelism"))
    rf = res.get("roofline")
    if rf is not None:
        o = pick(rf, ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "traffic",
                      "forward_kernel_ms", "forward_frac", "mfma_f64_frac", "measured_stream_read_GBs", "role_split_kernel_ms"))
        # the counter traffic comes from the committed rocprofv3 --pmc passes of these kernel sources (hash-guarded), never
        # from inside the timed process
        o["traffic_measured_in_this_run"] = False
        rp = rf.get("kernel_ms_rocprof_all_launches")
        if rp:
            o["kernel_ms_rocprof_avg"] = _r(rp.get("avg_ms"))
        out["roofline"] = o
    ss = res.get("strong_scaling")
    if ss is not None:   # configs[4] with the job fixed at `total_instances` (gather inside the timed region)
        out["strong_scaling"] = pick(ss, ("total_instances", "per_gpu_batch", "value", "ms_per_step", "backward_kernel_ms", "forward_kernel_ms",
                                          "gather_ms_per_step", "backward_hbm_frac"))
    cb = res.get("cpu_baseline")
    if cb is not None:
        o = pick(cb, ("value", "unit", "cores", "kind", "single_thread_sweeps_per_sec"))
        o["sample"] = cb.get("sample_short", "")
        if "sqp_iteration" in cb:
            o["sqp_iters_per_sec"] = _r(cb["sqp_iteration"].get("iters_per_sec"))
        if cb.get("build"):
            o["build"] = cb["build"].get("flags")
        rs = cb.get("reference_sources")
        if rs and "sweeps_per_sec" in rs:   # robotoc's own RiccatiRecursion behind the Eigen stand-in, one thread (see the detail record)
            o["reference_sources"] = {"sweeps_per_sec": _r(rs["sweeps_per_sec"]), "cores": 1, "build": "ref src -O2 +asserts, eager Eigen stand-in"}
        out["cpu_baseline"] = o
    sq = res.get("sqp_iteration")
    if sq is not None:
        out["sqp_iters_per_sec"] = _r(sq.get("iters_per_sec_per_gpu", 0.0) * res.get("n_gpus", 1))
        out["sqp_newton_iteration_ms"] = _r(sq.get("newton_iteration_ms"))
        out["sqp_phase_ms"] = {k: _r(v, 4) for k, v in sq.get("phase_ms", {}).items()}
        for key in ("roofline_condense", "roofline_expand"):
            if key in sq:
                out[key] = pick(sq[key], ("kernel_ms", "frac", "algorithmic_bytes_per_launch", "traffic"))
        if (sq.get("condense_register") or {}).get("role_split_ms"):
            out["condense_role_split_ms"] = _r(sq["condense_register"]["role_split_ms"], 4)   # RTOC_OPT_CONDENSE_REGISTER = 0 in the same loop
        cl = sq.get("closed_loop_constrained_trot", {}).get("batch")
        if cl:
            out["closed_loop_update_solution_ms"] = _r(cl.get("update_solution_ms"))
    oc = res.get("other_configs")
    if oc:
        o = {}
        for name, e in oc.items():
            c = pick(e, ("batch", "backward_ms", "forward_ms", "sweeps_per_sec", "single_instance_sweep_ms",
                         "single_instance_sweep_scan_ms", "sqp_iters_per_sec", "update_solution_iters_per_sec_batch256"))
            r2 = e.get("roofline")
            if r2:
                c["bound"] = r2.get("bound")
                c["hbm_frac"] = _r(r2["hbm"]["frac"], 4)
                c["mfma_f64_frac"] = _r(r2["mfma_f64"]["frac"], 4)
            if "roofline_condense" in e:
                c["condense_frac"] = _r(e["roofline_condense"].get("frac"), 4)
            if e.get("solve"):
                c["solve"] = pick(e["solve"], ("iterations", "converged", "final_kkt", "ms_per_iteration"))
            # this repo's own hop with the 2 x 17 wrench-cone rows, a fixed number of Gauss-Newton iterations without line search
            # (a timing; the reference's example of the configuration is icub_jump_sto_example_N130 above / below)
            sv = ((e.get("closed_loop") or {}).get("single_instance") or {}).get("solve")
            if sv:
                c["wrench_cone_hop_fixed_iterations"] = pick(sv, ("iterations", "final_kkt", "ms_per_iteration"))
            o[name] = c
        out["other_configs"] = o
    out["status_nonzero_instances"] = res.get("status_nonzero_instances", 0) + (sq or {}).get("status_nonzero_instances", 0)
    for k in ("rccl_gather_ok", "rccl_gather_c_abi_ok", "gather_backend", "dry_run", "gather_ok"):
        if k in res:
            out[k] = res[k]
    out["detail"] = res.get("detail_file")
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:   # never let the line outgrow the driver's tail again: shed the optional blocks
        for k in ("other_configs", "sqp_phase_ms", "roofline_expand", "roofline_condense"):
            out.pop(k, None)
            line = json.dumps(out, allow_nan=False, separators=(",", ":"))
            if len(line) <= COMPACT_LIMIT:
                break
    assert len(line) <= COMPACT_LIMIT, len(line)
    return line


def _finite(o):
    """NaN / inf -> None, so that the detail file is strict JSON too."""
    if isinstance(o, float):
        return o if np.isfinite(o) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return _finite(o.item())
    return o


def emit(res, detail_path):
    """Write the full record to `detail_path` (best effort) and print the compact line as the LAST line of stdout."""
    res = _finite(res)
    try:
        os.makedirs(os.path.dirname(detail_path), exist_ok=True)
        with open(detail_path, "w") as f:
            json.dump(res, f, allow_nan=False)
            f.write("\n")
        res["detail_file"] = os.path.relpath(detail_path, ROOT)
    except OSError as e:
        res["detail_file"] = "not written: %r" % (e,)
    sys.stdout.flush()
    print(compact_line(res), flush=True)


def sqp_single_instance(dims, grids, device):
    """SQP hot path of ONE OCP (the regime of a robotoc::OCPSolver call): per-phase HIP-event times with the
    serial recursions and with both recursions as horizon scans (RTOC_OPT_BACKWARD_SCAN)."""
    from robotoc_amd import capi, problems as pr
    from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_DX0, BUF_KKT, joint_limit_rows
    out = {}
    for mode in ("serial", "scan", "scan_graph"):
        c1 = capi.Context(dims, len(grids), 1, device)
        L1 = c1.L
        c1.set_grid(grids)
        c1.set_constraint_rows(joint_limit_rows(dims))
        c1.set_friction_cones(4, 3)
        c1.set_backward_scan(mode.startswith("scan"))
        c1.set_graph(mode.endswith("graph"))  # RTOC_OPT_GRAPH: the iteration replayed from a captured hipGraph
        kkt, cdd = pr.make_precondense_batch(L1, grids, 1)
        con = pr.make_constraint_batch(L1, grids, 1)
        c1.upload(BUF_CONE, pr.make_cone_batch(L1, grids, 1, 4))
        c1.upload(BUF_DX0, pr.make_dx0(L1, 1))
        ph = {"condense": 2, "backward": 0, "forward": 1, "expand": 3, "update": 5}
        acc = {k: 0.0 for k in ph}
        whole = 0.0
        nrep = 3
        for rep in range(nrep + 1):
            c1.upload(BUF_KKT, kkt)
            c1.upload(BUF_CDD, cdd)
            c1.upload(BUF_CON, con)
            for name in ("condense", "backward", "forward", "expand", "update"):
                ms = c1.time_phase(ph[name], 1)
                if rep > 0:
                    acc[name] += ms / nrep
        for rep in range(nrep + 3):
            c1.upload(BUF_KKT, kkt)
            c1.upload(BUF_CDD, cdd)
            c1.upload(BUF_CON, con)
            ms = c1.time_phase(6, 1)  # rtoc_newton_iteration: the whole iteration as one launch sequence
            if rep > 2:  # (graph mode: call 1 warms up, call 2 captures, then replays)
                whole += ms / nrep
        ok = int((c1.status() != 0).sum()) == 0
        c1.close()
        out[mode] = {"ms": acc, "newton_iteration_ms": whole, "iters_per_sec": 1e3 / whole, "status_ok": ok}
    return out


def closed_loop_trot(local_rank, batch, iters=30, timed=10):
    """OCPSolver::updateSolution of BASELINE configs[1] with NOTHING on the host: ANYmal trot (47 grid points: 2 lifts,
    2 touch-downs with switching constraints), ConfigurationSpaceCost, the Constraints object of examples/anymal/trot.cpp
    (six joint-limit components + FrictionCone), a distinct initial state per instance.  One iteration =
    rtoc_contact_update_solution: cost, inequality rows, state equation on SE(3), RNEA + derivatives, switching constraints,
    KKT error, condensation, Riccati sweep, expansion, fraction-to-boundary steps, update on the manifold."""
    from robotoc_amd import capi, problems as pr, robot_model as rm
    from robotoc_amd.grid import ANYMAL_Q_STANDING, ANYMAL_TROT_IMPACT_MASKS, ANYMAL_TROT_PHASE_MASKS, contact_masks
    from robotoc_amd.types import BUF_SOL, GRID_IMPACT, Records, joint_limit_rows
    m = rm.load_named("anymal")
    dims, grids, _ = pr.config_anymal_trot()
    n, nv, nq, nu = len(grids), m.nv, m.nq, 12
    qs = np.array(ANYMAL_Q_STANDING, dtype=float)
    masks = contact_masks(grids, ANYMAL_TROT_PHASE_MASKS, ANYMAL_TROT_IMPACT_MASKS)
    feet = np.array([m.frame_placement(qs, c)[1] for c in range(4)])
    pos = np.tile(feet[None], (n, 1, 1))
    impacts = [i for i, g in enumerate(grids) if g.type == GRID_IMPACT]
    pos[impacts[0]:, [1, 2], 0] += 0.05   # each swing foot touches down 5 cm ahead
    pos[impacts[1]:, [0, 3], 0] += 0.05
    c = capi.Context(dims, n, batch, local_rank)
    c.set_grid(grids)
    c.set_robot_model(m)
    c.set_contact_schedule(masks, pos)
    c.set_constraint_rows(joint_limit_rows(dims))
    c.set_friction_cones(4, 3)
    c.set_impact_cones(False)
    c.set_constraint_bounds(np.concatenate([np.full(2 * nu, 1.2), np.full(2 * nu, 3.0), np.full(2 * nu, 17.34)]), 1.0e-3, 0.995)
    c.set_friction_coefficients(np.full(4, 0.2))
    q_ref = qs.copy()
    q_ref[0] += 0.15
    wq = np.concatenate([np.full(6, 10.0), np.full(12, 1.0)])
    c.set_configuration_cost(q_ref, np.zeros(nv), np.zeros(12), wq, np.full(nv, 1.0), np.full(nv, 1e-3), np.full(12, 1e-3), 10.0 * wq,
                             np.full(nv, 1.0), q_weight_impact=wq, v_weight_impact=np.full(nv, 1.0), dv_weight_impact=np.full(nv, 1e-3))
    rng = np.random.default_rng(99)
    x0 = np.tile(np.concatenate([qs, np.zeros(nv)]), (batch, 1))
    x0[:, :3] += 0.01 * rng.uniform(-1, 1, (batch, 3))           # a distinct initial state per instance
    x0[:, 7:nq] += 0.02 * rng.uniform(-1, 1, (batch, nq - 7))
    x0[:, nq:] = 0.05 * rng.uniform(-1, 1, (batch, nv))
    c.set_initial_state(x0)
    S = Records(c.L, "sol")
    sol = S.zeros(batch, n)
    S.f(sol, "q")[..., :nq] = x0[:, None, :nq]
    mass = sum(m.mass[i] for i in range(m.njoints))
    for i in range(n):   # gravity-compensating contact forces at the standing pose as the initial guess
        act = [k for k in range(4) if (int(masks[i]) >> k) & 1]
        if act and grids[i].type != GRID_IMPACT:
            S.f(sol, "f")[:, i, :3 * len(act)] = np.concatenate([m.frame_placement(qs, k)[0].T @ np.array([0.0, 0.0, 9.81 * mass / len(act)]) for k in act])
    c.upload(BUF_SOL, sol)
    c.contact_init_constraints()
    errs = np.array([c.contact_update_solution(0.995) for _ in range(iters)])
    converged = int((errs[-1] < 1e-6).sum())
    c.upload(BUF_SOL, sol)
    c.contact_init_constraints()
    c.contact_update_solution(0.995, want_kkt_error=False)
    c.sync()
    t0 = time.perf_counter()
    for _ in range(timed):
        c.contact_update_solution(0.995, want_kkt_error=False)
    c.sync()
    ms = (time.perf_counter() - t0) / timed * 1e3
    out = {"batch": batch, "grid_points": n, "update_solution_ms": ms, "iterations_per_sec": batch / ms * 1e3,
           "kkt_error_first_worst": float(errs[0].max()), "kkt_error_after_%d_iterations_worst" % iters: float(errs[-1].max()),
           "kkt_error_after_%d_iterations_median" % iters: float(np.median(errs[-1])),
           "instances_below_1e-6": converged, "status_ok": bool((c.status() == 0).all()),
           "scope": "the WHOLE OCPSolver::updateSolution on the device, rigid-body linearisation, cost and inequality rows included "
                    "(joint limits + friction cones, barrier 1e-3); wall clock around asynchronous launches, synchronised once"}
    c.close()
    return out


def closed_loop_icub(local_rank, batch, nv=32, iters=12, timed=10):
    """OCPSolver::updateSolution of BASELINE configs[3] with nothing on the host: iCub through stand - flight - touch-down of
    both soles (N = 30: lift grid, flight phase, impact grid with its 12-row switching constraint), ConfigurationSpaceCost, the
    six joint-limit components and ContactWrenchCone / ImpactWrenchCone (2 x 17 rows per grid point with both soles down), a
    distinct initial state per instance.  nv = 32: the reference URDF's iCub with the torso locked (robot_model.lock_joints;
    BASELINE.json names that size), nv = 35: the URDF as it is.  The plain Gauss-Newton iteration with these inequality rows is
    what the reference runs without its line search; it is TIMED here (every kernel of the iteration runs whatever the iterate),
    the KKT errors of the first `iters` iterations are reported beside it."""
    from robotoc_amd import capi, robot_model as rm
    from robotoc_amd.grid import ICUB_Q_STANDING, ContactSequence, Event, contact_masks, discretize
    from robotoc_amd.types import BUF_SOL, GRID_IMPACT, Records, icub_dims, joint_limit_rows
    m = rm.load_named("icub32" if nv == 32 else "icub")
    assert m.nv == nv
    nq, nu = m.nq, nv - 6
    qs = np.array(ICUB_Q_STANDING, dtype=float)
    if nv == 32:
        qs = np.delete(qs, [19, 20, 21])
    dims = icub_dims(nv, nc_max=(6 * nu + 34 + 7) & ~7)
    grids = discretize(30, 0.6, 0.0, ContactSequence([12, 0, 12], [Event("lift", 0.25), Event("impact", 0.36, impact_dimf=12)]))
    n = len(grids)
    masks = contact_masks(grids, [0b11, 0, 0b11], [0b11])
    place = [m.frame_placement(qs, c) for c in range(2)]
    pos = np.tile(np.array([p for _, p in place])[None], (n, 1, 1))
    rot = np.tile(np.array([R.reshape(9) for R, _ in place])[None], (n, 1, 1))
    c = capi.Context(dims, n, batch, local_rank)
    c.set_grid(grids)
    c.set_robot_model(m)
    c.set_contact_schedule(masks, pos, rot)
    c.set_constraint_rows(joint_limit_rows(dims))
    c.set_wrench_cones(2)
    c.set_constraint_bounds(np.concatenate([np.full(2 * nu, 2.5), np.full(2 * nu, 5.0), np.full(2 * nu, 60.0)]), 1.0e-3, 0.995)
    c.set_wrench_cone_params(np.array([[0.1, 0.05, 0.6], [0.1, 0.05, 0.6]]))
    wq = np.concatenate([np.full(6, 10.0), np.full(nu, 0.1)])
    c.set_configuration_cost(qs, np.zeros(nv), np.zeros(nu), wq, np.full(nv, 0.1), np.full(nv, 1e-3), np.full(nu, 1e-4), 10 * wq, np.full(nv, 0.1),
                             q_weight_impact=wq, v_weight_impact=np.full(nv, 0.1), dv_weight_impact=np.full(nv, 1e-3))
    rng = np.random.default_rng(101)
    x0 = np.tile(np.concatenate([qs, np.zeros(nv)]), (batch, 1))
    x0[:, nq:] = 0.02 * rng.uniform(-1, 1, (batch, nv))           # a distinct initial state per instance
    c.set_initial_state(x0)
    S = Records(c.L, "sol")
    sol = S.zeros(batch, n)
    S.f(sol, "q")[..., :nq] = x0[:, None, :nq]
    mass = sum(m.mass[i] for i in range(m.njoints))
    f0 = np.concatenate([np.concatenate([R.T @ np.array([0.0, 0.0, 9.81 * mass / 2]), np.zeros(3)]) for R, _ in place])
    for i in range(n):
        if masks[i] and grids[i].type != GRID_IMPACT:
            S.f(sol, "f")[:, i, :12] = f0
    c.upload(BUF_SOL, sol)
    c.contact_init_constraints()
    errs = np.array([c.contact_update_solution(0.995) for _ in range(iters)])
    c.upload(BUF_SOL, sol)
    c.contact_init_constraints()
    c.contact_update_solution(0.995, want_kkt_error=False)
    c.sync()
    t0 = time.perf_counter()
    for _ in range(timed):
        c.contact_update_solution(0.995, want_kkt_error=False)
    c.sync()
    ms = (time.perf_counter() - t0) / timed * 1e3
    final = float(errs[-1].max())
    out = {"batch": batch, "nv": nv, "grid_points": n, "update_solution_ms": ms,
           # what a solve of this problem looks like from here (no iterations/s for a history that does not converge)
           "solve": {"iterations": int(iters), "converged": bool(final < 1e-7), "final_kkt": final, "ms_per_iteration": ms},
           "kkt_error_worst_by_iteration": [float(e.max()) for e in errs], "status_ok": bool((c.status() == 0).all()),
           "inequality_rows_per_grid_point": 6 * nu + 34,
           "scope": "the WHOLE OCPSolver::updateSolution on the device (cost, joint limits + 2 x 17 wrench-cone rows, state equation on "
                    "SE(3), RNEA + derivatives, switching constraint, KKT error, condensation, Riccati sweep, expansion, steps, update); "
                    "Gauss-Newton iterations from the standing guess without the line search (timing; the KKT history is reported, not "
                    "claimed as converged: with the merit-backtracking line search the same problem reaches 1e-4 (nv 35) / 9e-2 (nv 32) in "
                    "150 iterations and stalls there, tools/icub_solve_probe.py); wall clock around asynchronous launches, synchronised once"}
    c.close()
    return out


def dry_run(args, rank, world):
    """The multi-rank plumbing of this file without a GPU: rendezvous (gloo), barrier + max-over-ranks timing, the
    all-gather of (dummy) direction records through robotoc_amd.sharding, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from robotoc_amd.sharding import gather_directions
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = 3
    local = torch.full((batch, 2, 4), float(rank), dtype=torch.float64)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    full = gather_directions(local, world * batch, world, rank)
    ok = full.shape[0] == world * batch and all(float(full[r * batch, 0, 0]) == float(r) for r in range(world))
    if rank == 0:
        emit({"metric": "riccati_sweeps_per_sec", "dry_run": True, "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "gather_ok": bool(ok), "value": None}, args.detail_out)
    if world > 1:
        dist.destroy_process_group()


def solve_icub_jump_sto_example(local_rank):
    """BASELINE configs[3] as the reference poses it (examples/icub/python/jump_sto.py; robotoc_amd.problems_jump.icub_jump_sto_solver):
    iCub nv = 35 on its two soles (surface contacts), two jumps, four switching times optimised, the example's cost weights, joint limits
    of its URDF, FrictionCone on the soles, minimum dwell times, N = 130, max_iter = 350 -- OCPSolver::solve of ONE OCP on the device."""
    from robotoc_amd import problems_jump as pj
    solver, x0, info = pj.icub_jump_sto_solver(batch=1, device=local_rank)
    t0 = time.perf_counter()
    st = solver.solve(0.0, x0)
    wall = time.perf_counter() - t0
    errs = np.array(st.kkt_error)
    out = {"batch": 1, "nv": 35, "N": info["N"], "grid_points": len(solver.grids),
           "solve": {"iterations": int(st.iter), "converged": bool(st.convergence), "final_kkt": float(errs[-1].max()),
                     "ms_per_iteration": wall * 1e3 / max(int(st.iter), 1)},
           "kkt_error_first": float(errs[0].max()), "mesh_refinements_at": list(st.mesh_refinement_iter),
           "event_times_initial": [0.7, 0.95, 1.65, 1.9], "event_times_optimised": [float(v) for v in solver.event_times[0]],
           "status_ok": bool((solver.ctx.status() == 0).all()),
           "with_horizon_scan": None,
           "scope": "the reference example's OCP and solver options; every iteration (evalKKT incl. RNEA + derivatives of the 35-dof model, "
                    "condensation, STO Riccati recursion, expansion, step sizes, update) on the device, schedule and mesh refinement on the host"}
    solver.close()
    # the same solve with the backward / forward recursions as horizon scans (RTOC_OPT_BACKWARD_SCAN: stage parallelism for ONE OCP)
    solver, x0, info = pj.icub_jump_sto_solver(batch=1, device=local_rank, horizon_scan="on")
    t0 = time.perf_counter()
    st = solver.solve(0.0, x0)
    wall = time.perf_counter() - t0
    out["with_horizon_scan"] = {"iterations": int(st.iter), "converged": bool(st.convergence), "final_kkt": float(np.array(st.kkt_error)[-1].max()),
                                "ms_per_iteration": wall * 1e3 / max(int(st.iter), 1)}
    solver.close()
    # throughput: the same OCP in 256 instances with distinct initial states, OCPSolver::updateSolution of the whole batch timed
    b2 = 256
    solver, x0, info = pj.icub_jump_sto_solver(batch=b2, device=local_rank, x0_noise=0.01)
    c = solver.ctx
    c.set_initial_state(x0)
    solver.init_constraints()
    c.sto_set_regularization(1.0e30)
    errs = [c.contact_update_solution(0.995) for _ in range(3)]
    c.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        c.contact_update_solution(0.995, want_kkt_error=False)
    c.sync()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out["batch_throughput"] = {"batch": b2, "grid_points": len(solver.grids), "update_solution_ms": ms, "iterations_per_sec": b2 / ms * 1e3,
                               "kkt_error_worst_first_iterations": [float(np.max(e)) for e in errs], "status_ok": bool((c.status() == 0).all())}
    out["update_solution_iters_per_sec_batch256"] = b2 / ms * 1e3
    solver.close()
    return out


def closed_loop_jump_sto(local_rank, timed=10):
    """OCPSolver::solve of BASELINE configs[2] with nothing of the iteration on the host: stand - flight - stand, both switching
    times optimised (SwitchingTimeOptimization on the device: dwell-time rows, per-instance event times and time steps),
    ConfigurationSpaceCost, six joint-limit components + FrictionCone; mesh refinement on the host between iterations."""
    from robotoc_amd import problems_jump as pj
    out = {}
    for label, b2 in (("single_instance", 1), ("batch", 1024)):
        solver, x0, info = pj.anymal_jump_sto_solver(batch=b2, device=local_rank, x0_noise=0.0 if b2 == 1 else 0.03)
        c = solver.ctx
        t0 = time.perf_counter()
        st = solver.solve(0.0, x0)
        solve_s = time.perf_counter() - t0
        errs = np.array(st.kkt_error)
        ts = solver.event_times
        c.contact_update_solution(0.995, want_kkt_error=False)
        c.sync()
        t0 = time.perf_counter()
        for _ in range(timed):
            c.contact_update_solution(0.995, want_kkt_error=False)
        c.sync()
        ms = (time.perf_counter() - t0) / timed * 1e3
        out[label] = {"batch": b2, "grid_points": len(solver.grids), "update_solution_ms": ms, "iterations_per_sec": b2 / ms * 1e3,
                      "solve_iterations": st.iter, "solve_converged": bool(st.convergence), "solve_wall_s": solve_s,
                      "mesh_refinements_at": st.mesh_refinement_iter, "kkt_error_first_worst": float(errs[0].max()),
                      "kkt_error_last_worst": float(errs[-1].max()), "event_times_initial": [0.31, 0.51],
                      "event_times_optimised_instance0": [float(v) for v in ts[0]],
                      "event_times_optimised_spread": [float(v) for v in (ts.max(axis=0) - ts.min(axis=0))],
                      "status_ok": bool((c.status() == 0).all())}
        if b2 == 1:   # the same iteration with the backward recursion as a horizon scan (RTOC_OPT_BACKWARD_SCAN, DESIGN 3.6 (profiles/HISTORY.md 6b))
            c.set_backward_scan(True)
            c.contact_update_solution(0.995, want_kkt_error=False)
            c.sync()
            t0 = time.perf_counter()
            for _ in range(timed):
                c.contact_update_solution(0.995, want_kkt_error=False)
            c.sync()
            out[label]["update_solution_ms_with_horizon_scan"] = (time.perf_counter() - t0) / timed * 1e3
        solver.close()
    out["scope"] = ("the WHOLE OCPSolver::updateSolution incl. its switching-time half on the device (correctTimeSteps, sto_.evalKKT, "
                    "computeStepSizes, integrateSolution; cost, joint limits + friction cones, state equation, RNEA + derivatives, switching "
                    "constraint, condensation, STO Riccati recursion, expansion, update); OCPSolver::solve's regularisation schedule, "
                    "convergence test and mesh refinement (with solution interpolation) on the host; wall clock around asynchronous launches")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)   # (the clock settles over the first ~8 launches: profiles/r03_dvfs_probe.txt)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="instances per GPU")
    ap.add_argument("--waves", type=int, default=0, help="backward-kernel waves per instance (0=default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sqp", action="store_true", help="skip the SQP-iteration phase timing")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json configurations")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="file the full (20+ KB) record goes to; stdout carries only the compact line")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / gather plumbing only, no GPU work (tests/test_bench_launcher.py)")
    args = ap.parse_args()

    # `python bench.py --gpus N` launches its N ranks itself (one process per GPU under torch.distributed.run); when the
    # driver has already done so (WORLD_SIZE in the environment) this process IS one of the ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise RuntimeError("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("RTOC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0  # functional test of the multi-process path on a single-GPU box
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # "nccl" is RCCL on ROCm; RTOC_BENCH_BACKEND=gloo only for the single-GPU functional test of
        # this multi-process path (RCCL refuses two ranks on one device)
        dist.init_process_group(os.environ.get("RTOC_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    from robotoc_amd import capi, problems as pr
    from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, BUF_DIR, BUF_DX0, BUF_KKT, joint_limit_rows

    dims, grids, _ = pr.config_anymal_trot()
    batch = args.batch
    n = len(grids)
    ctx = capi.Context(dims, n, batch, local_rank)
    L = ctx.L
    ctx.set_grid(grids)
    if args.waves:
        ctx.set_backward_waves(args.waves)
    # a stream of this process's own (torch's default stream has the null handle, which rtoc_set_stream reads as "the
    # context's own stream"): the context launches on it and the torch events below are recorded on it
    stream = torch.cuda.Stream(device=dev)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)

    def dev_records(which):
        return torch.zeros((batch, n, getattr(L, which).stride), dtype=torch.float64, device=dev)

    # every instance of every rank is a different problem: the generator streams are keyed by rank
    kkt_t = pr.make_kkt_batch_unique(L, grids, batch, seed=rank, backend="torch", device=dev, out=dev_records("kkt"))
    dx0_t = pr.make_dx0_unique(L, batch, seed=rank, backend="torch", device=dev).contiguous()
    dir_t = dev_records("dir")  # lives in a torch tensor so that torch.distributed (RCCL) can gather it
    ctx.bind(BUF_KKT, kkt_t.data_ptr())
    ctx.bind(BUF_DX0, dx0_t.data_ptr())
    ctx.bind(BUF_DIR, dir_t.data_ptr())
    torch.cuda.synchronize()

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
    # HIP events on the launch stream (the stream handed to ctx.set_stream above) INSIDE the timed
    # region, so that the kernel times reported under `roofline` are those of exactly the launches `ms_per_step` covers
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        ctx.riccati_backward()
        ev[k][1].record(stream)
        ctx.riccati_forward()
        ev[k][2].record(stream)
    sync_all()
    dt = time.perf_counter() - t0
    ms_b = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    ms_f = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    bad = int((ctx.status() != 0).sum())
    # the role-split kernel (the default up to round 4; what RTOC_OPT_BACKWARD_REGISTER = 0 runs) on the same records, for the record
    ms_b_rs = None
    if args.waves == 0:
        from robotoc_amd.types import OPT_BACKWARD_REGISTER
        if ctx.get_option(OPT_BACKWARD_REGISTER):
            ctx.set_backward_register(False)
            ctx.time_phase(0, 3)
            ms_b_rs = ctx.time_phase(0, 10)
            ctx.set_backward_register(True)
            bad += int((ctx.status() != 0).sum())
    distinct = int(torch.unique(kkt_t[:, 0, L.kkt.off[2]]).numel())  # Qxx(0,0) of stage 0 of every instance

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
    # ... and what tuned streaming kernels reach (rtoc_bandwidth_probe: 16 B per lane, one front over memory, 8 loads in flight):
    # the better denominator -- torch's copy_ above is the runtime's copyBuffer kernel, which this box's HBM outruns
    try:
        stream_read_gbs, stream_copy_gbs = capi.bandwidth_probe(local_rank, 1 << 32)
    except Exception:
        stream_read_gbs = stream_copy_gbs = None

    gathered_ok, gathered_c_ok = None, None
    if world > 1:
        # the one exchange step: all-gather of the step directions, through torch.distributed (RCCL) and through
        # the C ABI's rtoc_gather_directions on an ncclComm_t of its own (what a C++ host calls)
        from robotoc_amd.sharding import RcclComm, gather_directions
        full = gather_directions(dir_t, world * batch, world, rank)
        torch.cuda.synchronize()
        gathered_ok = bool(full.shape[0] == world * batch and torch.isfinite(full).all().item())
        if dist.get_backend() == "nccl":
            comm = RcclComm(world, rank)
            out_c = torch.full((world,) + tuple(dir_t.shape), float("nan"), dtype=torch.float64, device=dev)
            ctx.gather_directions(comm.handle, out_c.data_ptr())
            ctx.sync()
            torch.cuda.synchronize()
            gathered_c_ok = bool(torch.equal(out_c.reshape(full.shape), full))
            comm.close()
        del full

    # ---- BASELINE configs[4] read as STRONG scaling: the SAME 4096 instances (args.batch in total) sharded over the ranks --
    #      args.batch / world per GPU -- one step = backward + forward sweep of the shard + the all-gather of the step
    #      directions (the one exchange step, INSIDE the timed region); barrier + synchronize on both sides, max over ranks.
    #      At world = 1 it is the headline measurement itself (no gather).  The weak-scaling line above stays the contract's
    #      `value`; this block is what the same node does when the job does not grow with it.
    strong = None
    if world > 1:
        from robotoc_amd.sharding import gather_directions
        per = max(1, batch // world)
        cs = capi.Context(dims, n, per, local_rank)
        cs.set_grid(grids)
        cs.set_stream(stream.cuda_stream)
        ric_s = torch.zeros((per, n, L.ric.stride), dtype=torch.float64, device=dev)
        for b_, t_ in ((BUF_KKT, kkt_t), (BUF_DX0, dx0_t), (BUF_DIR, dir_t)):   # the first `per` instances of this rank's records
            cs.bind(b_, t_.data_ptr())
        from robotoc_amd.types import BUF_RIC
        cs.bind(BUF_RIC, ric_s.data_ptr())
        shard = dir_t[:per]

        def strong_step():
            cs.riccati_backward()
            cs.riccati_forward()
            stream.synchronize()   # the collective runs on torch's stream: the directions of this step are complete
            return gather_directions(shard, world * per, world, rank)
        for _ in range(max(2, args.warmup // 2)):
            strong_step()
        sync_all()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        t0s = time.perf_counter()
        for k in range(args.steps):
            evs[k][0].record(stream)
            cs.riccati_backward()
            evs[k][1].record(stream)
            cs.riccati_forward()
            evs[k][2].record(stream)
            stream.synchronize()
            full_s = gather_directions(shard, world * per, world, rank)
        sync_all()
        dts = time.perf_counter() - t0s
        tts = torch.tensor([dts], dtype=torch.float64, device="cuda")
        dist.all_reduce(tts, op=dist.ReduceOp.MAX)
        dts = float(tts.item())
        ms_bs = sum(e[0].elapsed_time(e[1]) for e in evs) / args.steps
        ms_fs = sum(e[1].elapsed_time(e[2]) for e in evs) / args.steps
        strong = {"scaling": "strong", "total_instances": world * per, "per_gpu_batch": per, "n_gpus": world,
                  "value": world * per * args.steps / dts, "unit": "sweeps/s", "ms_per_step": dts / args.steps * 1e3,
                  "backward_kernel_ms": ms_bs, "forward_kernel_ms": ms_fs,
                  "gather_ms_per_step": max(0.0, dts / args.steps * 1e3 - ms_bs - ms_fs),
                  "gather_bytes_per_rank": int(shard.numel() * 8), "gather_inside_timed_region": True,
                  "backward_hbm_frac": algorithmic_bytes(L, grids, per, "backward") / (ms_bs * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "gathered_ok": bool(full_s.shape[0] == world * per and torch.isfinite(full_s).all().item()),
                  "status_nonzero_instances": int((cs.status() != 0).sum()),
                  "note": "the same %d instances whatever the rank count (BASELINE configs[4]); at %d per GPU the backward kernel has %.1f "
                          "waves per CU -- one dependency chain per SIMD or fewer, 10 us per grid point (DESIGN 4)" % (
                              world * per, per, per / 256.0)}
        del full_s
        cs.close()

    # ---- SQP-iteration hot path on pre-condensation stage data of `batch` distinct instances:
    #      rtoc_newton_iteration (KKT error -> condense -> backward -> forward -> expand -> step sizes ->
    #      convergence mask -> slack/dual update) timed end to end with HIP events on the launch stream, and
    #      phase by phase; the device-to-device restore of the in-place-mutated records is untimed ----
    sqp = None
    if not args.no_sqp:
        del kkt_t
        rows = joint_limit_rows(dims)
        ctx.set_constraint_rows(rows)
        ctx.set_friction_cones(4, 3)
        kkt0, cdd0 = pr.make_precondense_batch_unique(L, grids, batch, seed=rank, backend="torch", device=dev,
                                                      out=(dev_records("kkt"), dev_records("cdd")))
        con0 = pr.make_constraint_batch_unique(L, grids, batch, seed=rank, backend="torch", device=dev,
                                               out=dev_records("con"))
        cone_t = pr.make_cone_batch_unique(L, grids, batch, 4, seed=rank, backend="torch", device=dev).contiguous()
        kkt_w, cdd_w, con_w = torch.empty_like(kkt0), torch.empty_like(cdd0), torch.empty_like(con0)
        for b_, t_ in ((BUF_KKT, kkt_w), (BUF_CDD, cdd_w), (BUF_CON, con_w), (BUF_CONE, cone_t)):
            ctx.bind(b_, t_.data_ptr())

        def restore():
            kkt_w.copy_(kkt0)
            cdd_w.copy_(cdd0)
            con_w.copy_(con0)
            torch.cuda.synchronize()
        ph = {"condense": 2, "backward": 0, "forward": 1, "expand": 3, "update": 5}
        acc = {k: 0.0 for k in ph}
        nrep = 3
        for rep in range(nrep + 1):
            restore()
            for name in ("condense", "backward", "forward", "expand", "update"):
                ms = ctx.time_phase(ph[name], 1)
                if rep > 0:
                    acc[name] += ms / nrep
        whole, wall = 0.0, 0.0
        for rep in range(nrep + 1):
            restore()
            w0 = time.perf_counter()
            ms = ctx.time_phase(6, 1)  # synchronises on its closing event
            w1 = time.perf_counter() - w0
            if rep > 0:
                whole += ms / nrep
                wall += w1 * 1e3 / nrep
        bad_sqp = int((ctx.status() != 0).sum())
        # the condensation pipeline that is NOT this shape's default (RTOC_OPT_CONDENSE_SPLIT: 0 = one kernel -- wave 0 assembles
        # MJtJinv, wave 1 condenses the cone rows, wave 2 stages the inputs --, 1 = mjtjinv_kernel + condense_kernel), beside it
        from robotoc_amd.types import OPT_CONDENSE_SPLIT
        split_default = ctx.get_option(OPT_CONDENSE_SPLIT)
        ctx.set_condense_split(not split_default)
        other_ms = 0.0
        for rep in range(nrep + 1):
            restore()
            ms = ctx.time_phase(ph["condense"], 1)
            if rep > 0:
                other_ms += ms / nrep
        bad_sqp += int((ctx.status() != 0).sum())
        ctx.set_condense_split(split_default)
        # the register-chained condensation kernel (condense_rv.hpp, RTOC_OPT_CONDENSE_REGISTER = 1: the default where it applies,
        # friction-cone rows condensed inside it) and the role-split one-kernel condensation (= 0), interleaved in one loop
        from robotoc_amd.types import OPT_CONDENSE_REGISTER
        register_default = bool(ctx.get_option(OPT_CONDENSE_REGISTER)) and not split_default
        reg_ms = {"role_split": [], "register": []}
        if not split_default:
            for rep in range(3):
                for key, opt in (("role_split", False), ("register", True)):
                    ctx.set_condense_register(opt)
                    restore()
                    reg_ms[key].append(ctx.time_phase(ph["condense"], 1))
            bad_sqp += int((ctx.status() != 0).sum())
            ctx.set_condense_register(register_default)
        k_rv = "condense_rv_kernel<18, 12, 12, 12, true> (contact grid points, friction-cone rows inside) + condense_kernel<.., SPLIT = false> (impact grid points)"
        k_fused, k_split = "condense_kernel<.., SPLIT = false> (one kernel)", "mjtjinv_kernel + condense_kernel<.., SPLIT = true>"
        t_fused = pmc_traffic("condense_kernel<18, 12, 12, 12, false>")
        # condense_rv_kernel's own counted bytes + the impact grid points' share of the role-split kernel's (its launch on those few grid
        # points is not the geometry the counter summary keeps)
        n_imp = sum(1 for g in grids[:-1] if g.type == 1)
        t_rv = (lambda a, b: a + b * n_imp / (len(grids) - 1) if a and b else None)(pmc_traffic("condense_rv_kernel<18, 12, 12, 12, true>"), t_fused)
        t_split = (lambda a, b: a + b if a and b else None)(pmc_traffic("mjtjinv_kernel<18, 12, 12, 12>"), pmc_traffic("condense_kernel<18, 12, 12, 12, true>"))
        cb = condense_bytes(L, grids, batch)
        crb = constraint_row_bytes(grids, batch, rows, 4, dims.nv)
        eb = expand_bytes(L, grids, batch, rows, 4)
        sqp = {"newton_iteration_ms": whole, "newton_iteration_wall_ms": wall,
               "iters_per_sec_per_gpu": batch / whole * 1e3, "phase_ms": acc, "phase_sum_ms": sum(acc.values()),
               "distinct_instances": batch,
               "roofline_condense": {"bound": "hbm", "achieved": cb / (acc["condense"] * 1e-3) / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": cb / (acc["condense"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes_per_launch": cb, "kernel_ms": acc["condense"],
                                     "kernels": k_split if split_default else (k_rv if register_default else k_fused), "RTOC_OPT_CONDENSE_SPLIT": split_default,
                                     "RTOC_OPT_CONDENSE_REGISTER": int(register_default),
                                     "traffic": t_split if split_default else (t_rv if register_default else t_fused),
                                     "constraint_row_bytes_per_launch": crb,
                                     "constraint_row_bytes_note": "the PDIPM rows' own data (cone Jacobians, slack / dual / residual / cmpl in, cond "
                                                                  "out: Constraints::condenseSlackAndDual reads and writes them too) -- NOT in "
                                                                  "algorithmic_bytes_per_launch (SURVEY 8d counts the contact-dynamics condensation); "
                                                                  "the counted traffic includes them"},
               "roofline_condense_other_pipeline": {"bound": "hbm", "kernel_ms": other_ms, "achieved": cb / (other_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                                    "unit": "GB/s", "frac": cb / (other_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": cb,
                                                    "kernels": k_fused if split_default else k_split, "RTOC_OPT_CONDENSE_SPLIT": int(not split_default),
                                                    "traffic": t_fused if split_default else t_split},
               "roofline_expand": {"bound": "hbm", "achieved": eb / (acc["expand"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": eb / (acc["expand"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "algorithmic_bytes_per_launch": eb, "kernels": "expand_kernel + cone_expand_kernel",
                                   "traffic": pmc_traffic("expand_kernel")},
               "condense_register": {"register_ms": min(reg_ms["register"]) if reg_ms["register"] else None,
                                     "role_split_ms": min(reg_ms["role_split"]) if reg_ms["role_split"] else None,
                                     "note": "RTOC_OPT_CONDENSE_REGISTER = 1 | 0 timed alternately in one loop (the second timing in a process runs at a "
                                             "higher clock); profiles/r06_condense_register.txt has the same with and without rows"},
               "single_instance": sqp_single_instance(dims, grids, local_rank) if rank == 0 else None,
               "status_nonzero_instances": bad_sqp,
               "scope": "hot path downstream of the Pinocchio linearisation: KKT error, PDIPM condensation of the "
                        "72 joint-limit and 20 friction-cone rows + computeMJtJinv + condenseContact/ImpactDynamics + "
                        "Riccati backward/forward + expandContactDynamics primal/dual + PDIPM expansion, "
                        "fraction-to-boundary step sizes, convergence mask and slack/dual update; linearisation, "
                        "cost and the manifold update of q are CPU-side and excluded"}
        # ---- the rigid-body linearisation upstream of the condensation: ANYmal model table, trot contact
        #      schedule, a distinct random (q, v, a, f, u, beta, mu) per grid point generated in HBM ----
        if rank == 0:
            from robotoc_amd import robot_model as rm
            from robotoc_amd.types import BUF_SOL
            model = rm.load_named("anymal")
            ctx.set_robot_model(model)
            masks, flip = [], False
            for g in grids:
                masks.append(0b1111 if g.dimf == 12 else 0 if g.dimf == 0 else (0b0110 if flip else 0b1001))
                flip = flip != (g.dimf == 6)
            ctx.set_contact_schedule(np.array(masks, dtype=np.uint32), np.zeros((len(grids), 4, 3)))
            gen = torch.Generator(device=dev)
            gen.manual_seed(4242)
            sol_t = torch.zeros(batch, len(grids), L.sol.stride, dtype=torch.float64, device=dev)
            o = L.sol.off
            U = lambda n, s: s * (2.0 * torch.rand(batch, len(grids), n, dtype=torch.float64, device=dev, generator=gen) - 1.0)
            sol_t[:, :, o[0]:o[0] + 19] = U(19, 0.8)
            quat = torch.randn(batch, len(grids), 4, dtype=torch.float64, device=dev, generator=gen)
            sol_t[:, :, o[0] + 3:o[0] + 7] = quat / quat.norm(dim=-1, keepdim=True)
            for fld, n, sc in ((1, 18, 0.8), (2, 18, 0.8), (3, 12, 5.0), (4, 12, 20.0), (7, 18, 1.0), (8, 12, 1.0), (9, 6, 1.0)):
                sol_t[:, :, o[fld]:o[fld] + n] = U(n, sc)
            ctx.bind(BUF_SOL, sol_t.data_ptr())
            restore()
            ctx.time_phase(8, 1)
            lin_ms = min(ctx.time_phase(8, 1) for _ in range(3))
            lin_ms0 = min(ctx.time_phase(7, 1) for _ in range(3))
            nvm, nfm = 18, 12
            per_point = 8 * ((19 + 18 + 18 + 12 + 12 + 18 + 12 + 6)            # q, v, a, u, f, beta, mu, nu_passive in
                             + nvm * nvm + (nvm + nfm) * 2 * nvm + nfm * nvm + (nvm + nfm)  # dIDda, dIDCdqv, dCda, IDC out
                             + 2 * (2 * nvm + nvm + nfm + 12) + 6)              # lx, la, lf, lu read-modify-write, lu_passive
            lb = per_point * batch * (len(grids) - 1)
            points = batch * (len(grids) - 1)
            fpp, fsrc = linearize_flops()
            tf = fpp * points / (lin_ms0 * 1e-3) / 1e12 if fpp else None
            sqp["linearize"] = {"ms": lin_ms, "ms_without_multiplier_terms": lin_ms0, "grid_points": points,
                                "ns_per_grid_point": lin_ms * 1e6 / points,
                                # the roof of this kernel pair is the fp64 vector unit, not HBM: executed fp64 flops per grid point
                                # (counter pass) / time against the 78.6 TFLOP/s vector peak.  The counter pass runs the call
                                # without the multiplier terms, so ms_without_multiplier_terms is the time it is divided by.
                                "roofline": {"bound": "valu_f64", "achieved": tf, "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                             "frac": tf / VALU_F64_PEAK_TFLOPS if tf else None, "flops_per_grid_point": fpp,
                                             "flops_source": fsrc,
                                             "hbm": {"algorithmic_bytes_per_launch": lb, "achieved_GBps": lb / (lin_ms * 1e-3) / 1e9,
                                                     "frac_of_hbm_peak": lb / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                             "note": "one wave per grid point, one lane per tangent direction behind a level-parallel "
                                                     "values pre-pass; profiles/%s_rocprof_summary.txt has the SQ counters "
                                                     "(VALU issue, LDS, waits)" % PROFILE_ROUND},
                                "scope": "linearizeContactDynamics / linearizeImpactDynamics incl. the multiplier terms; NOT part of "
                                         "newton_iteration_ms, which starts from pre-condensation records; part of closed_loop_constrained_trot",
                                "status_nonzero_instances": int((ctx.status() != 0).sum())}
            # the storage plan of the tangent walk (rtoc_robot_model_plan: LDS per wave -> waves per CU, passes and the bodies each visits)
            plans = {}
            for rname in ("anymal", "icub32", "icub"):
                pl, bodies = capi.robot_model_plan(rm.load_named(rname))
                plans[rname] = {"tree_levels": pl.nlevels, "forward_tangent_slots": pl.nbranch, "dofs_per_pass": pl.dofs_per_pass,
                                "passes": pl.npass, "bodies_visited_per_pass": [len(b) for b in bodies], "lds_bytes_per_wave": pl.lds_bytes,
                                "waves_per_cu_by_lds": (160 * 1024) // ((pl.lds_bytes + 1279) // 1280 * 1280)}
            sqp["linearize"]["walk_plan"] = plans
            del sol_t
        del kkt0, cdd0, con0, kkt_w, cdd_w, con_w, cone_t

    # ---- the other BASELINE.json configurations (parity-test cases; reported, not the headline):
    #      batch throughput with a roofline block each, and the single-instance sweep latency a robotoc
    #      OCPSolver call would see ----
    others = None
    if rank == 0 and not args.no_configs:
        others = {}
        cfgs = [("anymal_trot_N40", pr.config_anymal_trot, 0), ("anymal_jump_sto_N40", pr.config_anymal_jump_sto, 4096),
                ("icub_nv32_jump_N30", lambda: pr.config_icub_jump(nv=32), 1024),
                ("icub_nv35_jump_N30", lambda: pr.config_icub_jump(nv=35), 1024), ("iiwa14_unconstr_N20", pr.config_iiwa14, 4096)]
        for name, fn, nb in cfgs:
            d2, g2, info = fn()
            if name.startswith("icub"):   # room for the 2 x 17 wrench-cone rows behind the six joint-limit components
                from robotoc_amd.types import icub_dims
                d2 = icub_dims(d2.nv, nc_max=(6 * d2.nu + 34 + 7) & ~7)
            entry = {"stages": len(g2)}
            for label, b2 in (("single_instance", 1), ("batch", nb)):
                if b2 == 0:
                    continue
                c2 = capi.Context(d2, len(g2), b2, local_rank)
                L2 = c2.L
                c2.set_grid(g2)
                k2 = torch.zeros((b2, len(g2), L2.kkt.stride), dtype=torch.float64, device=dev)
                if name.startswith("iiwa"):
                    from robotoc_amd.types import Records
                    k1 = Records(L2, "kkt").zeros(1, len(g2))
                    pr.fill_unconstr_instance(L2, len(g2), k1[0], np.random.default_rng(1))
                    k2[...] = torch.from_numpy(k1).to(dev)
                    c2.bind(BUF_KKT, k2.data_ptr())
                    c2.set_unconstr_dense(True)       # the general kernels (timed first, for comparison) ...
                    c2.unconstr_backward(info["dt"])  # ... need the structured A, B materialised once
                else:
                    pr.make_kkt_batch_unique(L2, g2, b2, seed=7, backend="torch", device=dev, out=k2)
                    c2.bind(BUF_KKT, k2.data_ptr())
                    if name.startswith("icub"):
                        # A BOUND buffer of an iCub-size shape is re-checked on the device before every backward recursion
                        # (RTOC_OPT_FXX_STRUCTURE = 0: the register-wide kernel never loads the structured rows, so it cannot
                        # verify them itself).  These records are written once, here, and never again: checked once, explicitly,
                        # then asserted -- the documented way for a host that owns its records.
                        entry["fxx_structured"] = bool(c2.check_fxx_structure())
                        if entry["fxx_structured"]:
                            c2.set_fxx_structure(2)
                x2 = pr.make_dx0_unique(L2, b2, seed=7, backend="torch", device=dev).contiguous()
                c2.bind(BUF_DX0, x2.data_ptr())
                torch.cuda.synchronize()
                c2.time_phase(4, 2)
                # median of five measurements of three launches each: these configurations are timed once, behind whatever the clocks
                # made of the headline part (the same kernel reads 1.84 ms in tools/rv_bench.py and 1.98-2.31 ms here, launch to launch)
                mb = sorted(c2.time_phase(0, 3) for _ in range(5))[2]
                mf = sorted(c2.time_phase(1, 3) for _ in range(5))[2]
                ok = int((c2.status() != 0).sum()) == 0
                if name.startswith("iiwa"):
                    # the path rtoc_unconstr_backward / _forward take by default: the structured recursion (block adds of P+,
                    # unconstr_riccati.hpp); wall clock around asynchronous launches.  mb / mf above: the general kernels on
                    # materialised A, B (RTOC_OPT_UNCONSTR_DENSE), kept as `dense_*`.
                    c2.set_unconstr_dense(False)

                    def wall(fn, reps=100 if b2 == 1 else 30):
                        fn()
                        c2.sync()
                        t0 = time.perf_counter()
                        for _ in range(reps):
                            fn()
                        c2.sync()
                        return (time.perf_counter() - t0) / reps * 1e3
                    sb, sf = wall(lambda: c2.unconstr_backward(info["dt"])), wall(lambda: c2.unconstr_forward(info["dt"]))
                    ok = ok and int((c2.status() != 0).sum()) == 0
                    entry["dense_%s_backward_ms" % label], entry["dense_%s_forward_ms" % label] = mb, mf
                    mb, mf = sb, sf
                if b2 == 1:
                    entry["single_instance_sweep_ms"] = mb + mf
                    entry["single_instance_backward_ms"] = mb
                    # RTOC_OPT_BACKWARD_SCAN: both recursions as scans over the horizon (latency path); on grids with
                    # switching-time optimisation the backward recursion is "matrix scan + serial vector pass"
                    # (riccati_scan_sto.hpp) and the forward recursion stays the serial kernel
                    c2.set_unconstr_dense(True)   # (the scan works on the general elements)
                    c2.set_backward_scan(True)
                    c2.time_phase(4, 2)
                    ms, msf = c2.time_phase(0, 5), c2.time_phase(1, 5)
                    c2.set_backward_scan(False)
                    entry["single_instance_backward_scan_ms"] = ms
                    entry["single_instance_forward_scan_ms"] = msf
                    entry["single_instance_sweep_scan_ms"] = ms + msf
                    entry["forward_scan"] = not any(g.sto or g.sto_next for g in g2)
                    ok = ok and int((c2.status() != 0).sum()) == 0
                else:
                    ab, fl = algorithmic_bytes(L2, g2, b2, "backward"), backward_flops(L2, g2, b2)
                    if name.startswith("iiwa"):   # structured: Fxx / Fvu are never read (nor exist)
                        nx_, nv_ = L2.nx, d2.nv
                        ab = 8 * b2 * ((len(g2) - 1) * (2 * nx_ * nx_ + 2 * nx_ * nv_ + nv_ * nv_ + 3 * nx_ + 2 * nv_) + 2 * (nx_ * nx_ + nx_))
                    entry.update({"batch": b2, "distinct_instances": b2, "backward_ms": mb, "forward_ms": mf,
                                  "sweeps_per_sec": b2 / (mb + mf) * 1e3,
                                  "roofline": {"kernel": "riccati_backward", "kernel_ms": mb,
                                               "hbm": {"achieved": ab / mb / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                       "frac": ab / mb / 1e6 / HBM_PEAK_GBS,
                                                       "algorithmic_bytes_per_launch": ab},
                                               "mfma_f64": {"achieved": fl / mb / 1e9, "peak": F64_MFMA_PEAK_TFLOPS,
                                                            "unit": "TFLOP/s", "frac": fl / mb / 1e9 / F64_MFMA_PEAK_TFLOPS,
                                                            "algorithmic_flops_per_launch": fl},
                                               "bound": "hbm" if ab / (HBM_PEAK_GBS * 1e9) > fl / (F64_MFMA_PEAK_TFLOPS * 1e12)
                                               else "mfma"}})
                    if name.startswith("icub") or name.startswith("anymal_jump"):
                        # SQP iterations/s of this configuration (north_star names it for iCub too): one rtoc_newton_iteration
                        # on distinct pre-condensation records with the joint-limit rows and, for iCub (BASELINE configs[3]:
                        # 2 surface contacts), the 2 x 17 ContactWrenchCone / ImpactWrenchCone rows per grid point with both
                        # soles down (contact_wrench_cone.cpp:209-270)
                        from robotoc_amd.types import BUF_CDD, BUF_CON, BUF_CONE, joint_limit_rows
                        rows2 = joint_limit_rows(d2)
                        wrench2 = name.startswith("icub")
                        if len(rows2) + (34 if wrench2 else 0) <= d2.nc_max:
                            c2.set_constraint_rows(rows2)
                            if wrench2:
                                c2.set_wrench_cones(2)
                                cones2 = [capi.wrench_cone_matrix(0.1, 0.05, 0.6), capi.wrench_cone_matrix(0.09, 0.055, 0.7)]
                                cone2 = torch.from_numpy(pr.make_wrench_cone_batch(L2, g2, b2, 2, cones2)).to(dev).contiguous()
                                c2.bind(BUF_CONE, cone2.data_ptr())
                            kk = torch.zeros((b2, len(g2), L2.kkt.stride), dtype=torch.float64, device=dev)  # the generators fill fields, not padding
                            cc = torch.zeros((b2, len(g2), L2.cdd.stride), dtype=torch.float64, device=dev)
                            nn = torch.zeros((b2, len(g2), L2.con.stride), dtype=torch.float64, device=dev)
                            flagged, seed2 = -1, 0
                            # random stage data: a seed whose every instance stays positive definite through the iteration
                            # (the kernels have no data-dependent branches; a flagged instance costs the same)
                            for seed2 in (11, 12, 13, 14, 15):
                                pr.make_precondense_batch_unique(L2, g2, b2, seed=seed2, backend="torch", device=dev, out=(kk, cc))
                                pr.make_constraint_batch_unique(L2, g2, b2, seed=seed2, backend="torch", device=dev, out=nn)
                                kw, cw, nw = kk.clone(), cc.clone(), nn.clone()
                                for b_, t_ in ((BUF_KKT, kw), (BUF_CDD, cw), (BUF_CON, nw)):
                                    c2.bind(b_, t_.data_ptr())
                                torch.cuda.synchronize()
                                c2.clear_status()
                                c2.time_phase(6, 1)
                                flagged = int((c2.status() != 0).sum())
                                if flagged == 0:
                                    break
                            best = 1e9
                            for rep in range(3):
                                kw.copy_(kk), cw.copy_(cc), nw.copy_(nn)
                                torch.cuda.synchronize()
                                best = min(best, c2.time_phase(6, 1))
                            entry["sqp_newton_iteration_ms"] = best
                            entry["sqp_iters_per_sec"] = b2 / best * 1e3
                            entry["sqp_data_seed"] = seed2
                            entry["sqp_status_nonzero_instances"] = flagged
                            entry["sqp_inequality_rows"] = {"joint_limit_rows": len(rows2), "wrench_cone_rows_per_contact": 17 if wrench2 else 0,
                                                            "max_surface_contacts": 2 if wrench2 else 0}
                            # phase by phase, with the roofline blocks of the two HBM-bound phases
                            ph2 = {"condense": 2, "backward": 0, "forward": 1, "expand": 3, "update": 5}
                            acc2 = {k: 1e9 for k in ph2}
                            for rep in range(2):
                                kw.copy_(kk), cw.copy_(cc), nw.copy_(nn)
                                torch.cuda.synchronize()
                                for pn in ("condense", "backward", "forward", "expand", "update"):
                                    acc2[pn] = min(acc2[pn], c2.time_phase(ph2[pn], 1))
                            entry["sqp_phase_ms"] = acc2
                            cb2, eb2 = condense_bytes(L2, g2, b2), expand_bytes(L2, g2, b2, rows2, 2 if wrench2 else 0, wrench=wrench2)
                            for key, nbytes, pn in (("roofline_condense", cb2, "condense"), ("roofline_expand", eb2, "expand")):
                                entry[key] = {"bound": "hbm", "achieved": nbytes / (acc2[pn] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": nbytes / (acc2[pn] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes,
                                              "kernel_ms": acc2[pn], "traffic": None}
                            if wrench2 and b2 == 1024:   # the counter passes ran this iteration at this size (tools/pmc_driver.py)
                                tail = "<%d, %d, %d, %d" % (d2.nv, d2.nu, d2.nf_max, d2.ns_max)
                                parts = [pmc_traffic("mjtjinv_kernel" + tail), pmc_traffic("condense_kernel" + tail)]
                                entry["roofline_condense"]["traffic"] = sum(parts) if all(parts) else None
                                parts = [pmc_traffic("expand_kernel<%d," % d2.nv), pmc_traffic("wrench_expand_kernel<%d," % d2.nv)]
                                entry["roofline_expand"]["traffic"] = sum(parts) if all(parts) else None
                            if wrench2:
                                del cone2
                            c2.clear_status()
                            c2.bind(BUF_KKT, k2.data_ptr())
                            del kk, cc, nn, kw, cw, nw
                entry["status_ok"] = entry.get("status_ok", True) and ok
                c2.close()
                del k2, x2
            others[name] = entry
        # ---- BASELINE configuration 1 closed on the device: UnconstrOCPSolver::updateSolution (cost, state equation,
        #      rigid-body linearisation, condensation, Riccati sweep, expansion, solution update) per iteration ----
        try:
            from robotoc_amd import robot_model as rm
            from robotoc_amd.types import BUF_SOL, Records
            d2, g2, info = pr.config_iiwa14()
            m2 = rm.load_named("iiwa14")
            nv2, n2 = m2.nv, len(g2)
            rng2 = np.random.default_rng(77)
            loop = {}
            for label, b2 in (("single_instance", 1), ("batch", 4096)):
                c2 = capi.Context(d2, n2, b2, local_rank)
                c2.set_grid(g2)
                c2.set_robot_model(m2)
                c2.set_configuration_cost(rng2.uniform(-0.8, 0.8, nv2), np.zeros(nv2), np.zeros(nv2), np.full(nv2, 10.0), np.full(nv2, 0.1),
                                          np.full(nv2, 0.01), np.full(nv2, 0.001), np.full(nv2, 10.0), np.full(nv2, 0.1))
                x0 = np.concatenate([rng2.uniform(-0.5, 0.5, (b2, nv2)), np.zeros((b2, nv2))], axis=1)  # a distinct initial state per instance
                c2.set_initial_state(x0)
                S2 = Records(c2.L, "sol")
                sol2 = S2.zeros(b2, n2)
                S2.f(sol2, "q")[..., :nv2] = x0[:, None, :nv2]
                c2.upload(BUF_SOL, sol2)
                errs = [c2.unconstr_update_solution(info["dt"]).max() for _ in range(12)]   # solve: the iterates converge
                c2.upload(BUF_SOL, sol2)
                c2.unconstr_update_solution(info["dt"], want_kkt_error=False)
                c2.sync()
                t0 = time.perf_counter()
                reps = 20
                for _ in range(reps):
                    c2.unconstr_update_solution(info["dt"], want_kkt_error=False)
                c2.sync()
                ms = (time.perf_counter() - t0) / reps * 1e3
                loop[label] = {"batch": b2, "update_solution_ms": ms, "iterations_per_sec": b2 / ms * 1e3,
                               "kkt_error_after_12_iterations": float(errs[-1]), "kkt_error_first": float(errs[0]),
                               "status_ok": bool((c2.status() == 0).all())}
                c2.close()
            loop["scope"] = ("the WHOLE UnconstrOCPSolver::updateSolution on the device (ConfigurationSpaceCost, forward-Euler state "
                             "equation, RNEA + derivatives, condensation, Riccati sweep, expansion, update; no joint-limit rows); wall clock "
                             "around asynchronous launches, synchronised once")
            others["iiwa14_unconstr_N20"]["closed_loop"] = loop
        except Exception as e:  # the sweep numbers above stand on their own
            others["iiwa14_unconstr_N20"]["closed_loop"] = {"error": repr(e)}

    # ---- BASELINE configs[3] closed on the device: the iCub hop with joint limits and wrench cones, one instance and 1024 ----
    if others is not None:
        for name, nv_ in (("icub_nv32_jump_N30", 32), ("icub_nv35_jump_N30", 35)):
            if name in others:
                try:
                    others[name]["closed_loop"] = {"single_instance": closed_loop_icub(local_rank, 1, nv_), "batch": closed_loop_icub(local_rank, 1024, nv_)}
                except Exception as e:  # the sweep numbers above stand on their own
                    others[name]["closed_loop"] = {"error": repr(e)}

    # ---- BASELINE configs[2] closed on the device: the ANYmal jump with switching-time optimisation (examples/anymal/python/
    #      jump_sto.py at N = 40) solved by robotoc_amd.solver.OCPSolver -- per-instance switching times, mesh refinement ----
    if others is not None and "anymal_jump_sto_N40" in others:
        try:
            others["anymal_jump_sto_N40"]["closed_loop"] = closed_loop_jump_sto(local_rank)
        except Exception as e:  # the sweep numbers above stand on their own
            others["anymal_jump_sto_N40"]["closed_loop"] = {"error": repr(e)}

    # ---- BASELINE configs[3] as the reference example poses it, solved: one OCP, N = 130 ----
    if others is not None and rank == 0:
        try:
            others["icub_jump_sto_example_N130"] = solve_icub_jump_sto_example(local_rank)
        except Exception as e:
            others["icub_jump_sto_example_N130"] = {"error": repr(e)}

    if rank == 0:
        total_sweeps = world * batch * args.steps
        value = total_sweeps / dt
        bytes_b = algorithmic_bytes(L, grids, batch, "backward")
        bytes_f = algorithmic_bytes(L, grids, batch, "forward")
        fl_b = backward_flops(L, grids, batch)
        ach = bytes_b / (ms_b * 1e-3) / 1e9
        from robotoc_amd.types import OPT_BACKWARD_REGISTER
        kname = ("riccati_backward_rv_kernel" if args.waves == 0 and ctx.get_option(OPT_BACKWARD_REGISTER) else
                 "riccati_backward_rs4_kernel" if args.waves in (0, 8) else "riccati_backward_kernel")
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
                                   "switching constraints ns=6), %d DISTINCT OCP instances per GPU (randomised "
                                   "stage data and dx0), backward+forward Riccati sweep" % batch,
                       "per_gpu_batch": batch, "distinct_instances_per_gpu": distinct, "stages": len(grids),
                       "parallelism": "instances sharded, dp%d" % world, "backward_waves": args.waves},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(kname) if batch == PER_GPU_BATCH else None,
                         "traffic_source": traffic_state(),
                         "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": ach / copy_gbs,
                         "measured_copy_kernel": "torch Tensor.copy_ (runtime copyBuffer), 1 GiB read + write",
                         "measured_stream_read_GBs": stream_read_gbs, "measured_stream_copy_GBs": stream_copy_gbs,
                         "frac_of_measured_stream_read": ach / stream_read_gbs if stream_read_gbs else None,
                         "frac_of_measured_stream_copy": ach / stream_copy_gbs if stream_copy_gbs else None,
                         "measured_stream_kernel": "rtoc_bandwidth_probe: 4 GiB, 16 B / lane, 8 loads in flight per wave, best of 5",
                         "kernel": kname, "kernel_ms": ms_b,
                         "kernel_ms_source": "HIP events on the launch stream inside the timed sweep loop of this run",
                         "kernel_ms_rocprof_all_launches": rocprof_kernel_ms(kname) if batch == PER_GPU_BATCH else None,
                         "frac_at_rocprof_all_launch_average": (lambda r: (bytes_b / (r["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if r else None)(
                             rocprof_kernel_ms(kname) if batch == PER_GPU_BATCH else None),
                         "algorithmic_bytes_per_launch": bytes_b,
                         "role_split_kernel_ms": ms_b_rs,
                         "mfma_f64_achieved_TFLOPs": fl_b / (ms_b * 1e-3) / 1e12,
                         "mfma_f64_frac": fl_b / (ms_b * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                         "forward_kernel_ms": ms_f,
                         "forward_achieved": bytes_f / (ms_f * 1e-3) / 1e9,
                         "forward_frac": bytes_f / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "forward_algorithmic_bytes_per_launch": bytes_f,
                         "forward_traffic": pmc_traffic("riccati_forward_kernel") if batch == PER_GPU_BATCH else None},
            "status_nonzero_instances": bad,
        }
        if sqp is not None:
            res["sqp_iteration"] = sqp
            if world == 1:
                try:
                    res["sqp_iteration"]["closed_loop_constrained_trot"] = {"single_instance": closed_loop_trot(local_rank, 1),
                                                                            "batch": closed_loop_trot(local_rank, PER_GPU_BATCH)}
                except Exception as e:  # the hot-path numbers stand on their own
                    res["sqp_iteration"]["closed_loop_constrained_trot"] = {"error": repr(e)}
        if others is not None:
            res["other_configs"] = others
        if strong is not None:
            res["strong_scaling"] = strong
        if gathered_ok is not None:
            res["rccl_gather_ok"] = gathered_ok
            res["rccl_gather_c_abi_ok"] = gathered_c_ok   # null: gloo functional test (RCCL refuses two ranks on one device)
            res["gather_backend"] = dist.get_backend()
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(L, grids, dims)
        emit(res, args.detail_out)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
