#!/bin/bash
# rocprofv3 kernel statistics of the closed-loop constrained trot iteration (tools/closed_loop_bench.py, batch only):
# every kernel of rtoc_contact_update_solution with its share.  Output: gpurun_out/closed_loop_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/cl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cl -o cl -- python $R/tools/closed_loop_bench.py ${1:-4096} --batch-only > /tmp/cl.log 2>&1
python - <<PY > $R/gpurun_out/closed_loop_kernel_stats.txt
import csv, glob
f = glob.glob("/tmp/cl/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("# rocprofv3 --kernel-trace --stats -- python tools/closed_loop_bench.py ${1:-4096} --batch-only   (41 iterations of rtoc_contact_update_solution)")
print("# kernel, calls, total_ms, avg_us, percent")
for r in rows:
    print("%s, %s, %.3f, %.1f, %s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep -E "update_solution_ms" /tmp/cl.log
head -30 $R/gpurun_out/closed_loop_kernel_stats.txt
