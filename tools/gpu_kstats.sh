#!/bin/bash
# Kernel-trace statistics of tools/pmc_driver.py (no torch in the process): per-kernel average durations of
# every hot-path kernel at its bench.py size.  Output: gpurun_out/kstats/*_kernel_stats.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o k -- python $R/tools/pmc_driver.py ${1:-5} > $OUT/kstats.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/kstats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %4s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
