#!/bin/bash
# SQ counters of linearize_contact_dynamics_kernel (tools/linearize_bench.py, no torch in the process)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
D="python $R/tools/linearize_bench.py ${1:-1024}"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/prof_lin1 -o lin1 -- $D > $OUT/prof_lin1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/prof_lin2 -o lin2 -- $D > $OUT/prof_lin2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("prof_lin1", "prof_lin2"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "linearize" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("%-24s mean per dispatch %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
tail -2 $OUT/prof_lin1.log
