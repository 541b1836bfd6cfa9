#!/bin/bash
# PMC passes for the backward kernel (each --pmc set in its own run; no tracing domains mixed in)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
W=${WAVES:-0}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc1 -o p1 --output-format csv -- python $R/tools/quick_bench.py 4096 ${W} nosweep > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/pmc2 -o p2 --output-format csv -- python $R/tools/quick_bench.py 4096 ${W} nosweep > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc3 -o p3 --output-format csv -- python $R/tools/quick_bench.py 4096 ${W} nosweep > $OUT/pmc3.log 2>&1
find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 -name "*counter_collection.csv" | while read f; do python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "riccati" in k:
        print(k, {c: sum(x)/len(x) for c,x in v.items()}, "n=", len(next(iter(v.values()))))
PY
done
