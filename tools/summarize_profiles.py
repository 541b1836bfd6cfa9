"""Condense the rocprofv3 outputs of tools/gpu_round.sh (gpurun_out/) into the tracked summaries:
profiles/<tag>_rocprof_summary.txt and profiles/<tag>_traffic.json (+ bench json, host info)."""
import collections, csv, json, os, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(R, "gpurun_out")
# RTOC_PROFILE_OUT: write the summaries somewhere else than profiles/ (on the GPU box: into gpurun_out/summary, so that the raw
# traces, which exceed what gpurun copies back, can be deleted there)
DST = os.environ.get("RTOC_PROFILE_OUT", os.path.join(R, "profiles"))
os.makedirs(DST, exist_ok=True)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
note = sys.argv[2] if len(sys.argv) > 2 else ""
lines = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline   (%s)" % note,
         "# counter passes: rocprofv3 --pmc <set> -- python tools/pmc_driver.py 3   (tools/gpu_pmc2.sh; same kernels, same sizes, no torch in the process)",
         "# Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev"]
for r in csv.DictReader(open(os.path.join(OUT, "prof_stats", "stats_kernel_stats.csv"))):
    lines.append(", ".join(r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")))
# per launch geometry (the whole-bench trace mixes batch and single-instance launches of the same kernel)
trace = os.path.join(OUT, "prof_stats", "stats_kernel_trace.csv")
if os.path.exists(trace):
    # every launch with the idle time of the GPU in front of it (end of the previous kernel of ANY kind to this start): after
    # ~50 ms of idling the chip needs a few ms to clock up again, and the first kernels behind a host synchronisation run ~10 %
    # slower (tools/dvfs_probe.py) -- the geometry groups therefore also report the launches that ran back to back (gap < 50 us)
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    acc, b2b = collections.OrderedDict(), collections.OrderedDict()
    prev_end = None
    for r in rows:
        start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = None if prev_end is None else start - prev_end
        prev_end = max(prev_end or 0, end)
        if "rtoc::" not in r["Kernel_Name"] and "mask_converged" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
               int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]), r["LDS_Block_Size"], r["Scratch_Size"],
               r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])
        acc.setdefault(key, []).append((end - start) / 1e6)
        if gap is not None and gap < 50000:
            b2b.setdefault(key, []).append((end - start) / 1e6)
    # the dominant kernel at the headline batch, launch by launch in time order: duration [ms] and the kernel that ran before it
    seq, prev_name = [], "-"
    for r in rows:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rtoc::", "")
        if ("riccati_backward_rs4_kernel" in nm and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) >= 1024) or \
                ("riccati_backward_rv_kernel" in nm and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) >= 4096):
            seq.append("%.3f<%s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, prev_name.split("<")[0][:22]))
        prev_name = nm
    lines.append("# the dominant backward kernel (riccati_backward_rv_kernel, or rs4 where RTOC_OPT_BACKWARD_REGISTER is off) at the headline batch, every launch of the trace in time order: ms<kernel in front of it")
    for i in range(0, len(seq), 8):
        lines.append("#   " + "  ".join(seq[i:i + 8]))
    lines.append("# same trace grouped by launch geometry: kernel, workgroups(x), y*z, threads, LDS bytes, scratch bytes/lane, VGPR, AGPR, SGPR, calls, avg ms, min ms, max ms"
                 " | the launches that started < 50 us behind the previous kernel (back to back): calls, avg ms")
    for k, v in acc.items():
        w = b2b.get(k, [])
        lines.append("%s, %d, %d, %d, %s, %s, %s, %s, %s, n=%d, avg %.4f, min %.4f, max %.4f" % (k + (len(v), sum(v) / len(v), min(v), max(v)))
                     + (" | back to back n=%d, avg %.4f" % (len(w), sum(w) / len(w)) if w else " | back to back n=0"))
traffic = {}
# average duration of every kernel over ALL launches of its largest launch geometry in the kernel trace of the bench command
# (bench.py prints it beside its own event-timed kernel_ms: the trace mixes sweep-loop, in-iteration and other launches)
kernel_ms = {}
if os.path.exists(trace):
    for k, v in acc.items():
        w = b2b.get(k, [])
        cur = kernel_ms.get(k[0])
        if cur is None or k[1] * k[2] > cur["workgroups"]:
            kernel_ms[k[0]] = {"workgroups": k[1] * k[2], "calls": len(v), "avg_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v),
                               "back_to_back_calls": len(w), "back_to_back_avg_ms": (sum(w) / len(w)) if w else None}
for cname, d in (("FETCH_SIZE", "prof_fetch/fetch"), ("WRITE_SIZE", "prof_write/write")):
    acc = collections.OrderedDict()
    bygrid = {}
    for r in csv.DictReader(open(os.path.join(OUT, d + "_counter_collection.csv"))):
        if r["Counter_Name"] == cname:
            bygrid.setdefault(r["Kernel_Name"], {}).setdefault(int(float(r.get("Grid_Size", 0) or 0)), []).append(float(r["Counter_Value"]))
    for k, g in bygrid.items():   # a kernel launched with several geometries (condense_kernel: all grid points | the impact ones): the largest
        acc[k] = g[max(g)]
    lines.append("# rocprofv3 --pmc %s (own pass); mean per dispatch of the kernel's largest launch geometry, unit KB (rocprofv3 FETCH_SIZE/WRITE_SIZE)" % cname)
    for k, v in acc.items():
        m = sum(v) / len(v)
        lines.append("%s, %s, %.1f, n=%d" % (k, cname, m, len(v)))
        if "rtoc::" in k:
            short = k.split("(")[0].replace("void ", "")
            traffic.setdefault(short, {})["fetch_size_kb" if cname == "FETCH_SIZE" else "write_size_kb"] = m
for k, v in list(traffic.items()):
    v["hbm_bytes"] = (2.0 * v.get("fetch_size_kb", 0.0) + v.get("write_size_kb", 0.0)) * 1024.0
    v["note"] = "2*FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE (calibrated on a 295.7 MB torch fill: exact)"
# SQ counter passes (tools/gpu_round2.sh): mean per dispatch and kernel
for d in ("prof_sq1/sq1", "prof_sq2/sq2"):
    path = os.path.join(OUT, d + "_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "rtoc::" in r["Kernel_Name"]:
            acc.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), collections.OrderedDict()).setdefault(
                r["Counter_Name"], []).append(float(r["Counter_Value"]))
    lines.append("# rocprofv3 --pmc (own pass, %s): mean per dispatch" % d.split("/")[0])
    for k, cs in acc.items():
        lines.append("%s: %s n=%d" % (k, ", ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in cs.items()),
                                      len(next(iter(cs.values())))))
sys.path.insert(0, R)
import bench  # kernel_source_hash: bench.py refuses counter passes taken from other kernel sources
traffic["_kernel_source_hash"] = bench.kernel_source_hash()
traffic["_kernel_ms_rocprof"] = kernel_ms
open(os.path.join(DST, tag + "_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(DST, tag + "_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(OUT, "bench.json"), os.path.join(DST, tag + "_bench.json"))
shutil.copy(os.path.join(OUT, "host.txt"), os.path.join(DST, tag + "_host.txt"))
print("\n".join(lines[:6])); print(json.dumps(traffic, indent=1))
