#!/bin/bash
# PMC passes for the scan kernels, one ANYmal / iCub instance (each --pmc set in its own run; no tracing domains)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for cfg in ${CFGS:-anymal icub32}; do
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmcs1_$cfg -o p1 --output-format csv -- python $R/tools/scan_latency.py $cfg 1 10 > $OUT/pmcs1_$cfg.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/pmcs2_$cfg -o p2 --output-format csv -- python $R/tools/scan_latency.py $cfg 1 10 > $OUT/pmcs2_$cfg.log 2>&1
echo "== $cfg"
find $OUT/pmcs1_$cfg $OUT/pmcs2_$cfg -name "*counter_collection.csv" | while read f; do python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0].replace("void rtoc::","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "scan" in k:
        for c,x in sorted(v.items()):
            print("%s, %s, %.4g, n=%d" % (k, c, sum(x)/len(x), len(x)))
PY
done
done
