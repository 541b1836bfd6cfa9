#!/bin/bash
# usage: pmc_cmd.sh "<python command>"  -- three PMC passes, prints per-kernel means
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="$1"
rm -rf $OUT/pmcA $OUT/pmcB
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $OUT/pmcA -o a --output-format csv -- $CMD > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmcB -o b --output-format csv -- $CMD > $OUT/pmcB.log 2>&1
for f in $OUT/pmcA/a_counter_collection.csv $OUT/pmcB/b_counter_collection.csv; do python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "rtoc" in k:
        print(k, {c: "%.3g" % (sum(x)/len(x)) for c,x in v.items()})
PY
done
