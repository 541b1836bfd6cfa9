#!/bin/bash
# instruction-cache counters of the hot kernels (own PMC pass, no tracing domains)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf $OUT/pmcI
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/pmcI -o i --output-format csv -- python $R/tools/sqp_bench.py 4096 > $OUT/pmcI.log 2>&1
python3 - "$OUT/pmcI/i_counter_collection.csv" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "rtoc" in k:
        print(k, {c: "%.4g" % (sum(x)/len(x)) for c,x in v.items()})
PY
tail -2 $OUT/pmcI.log
