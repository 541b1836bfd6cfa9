#!/bin/bash
# instruction-cache behaviour of the backward kernel (own PMC pass)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/pmcIB -o i --output-format csv -- python $R/tools/quick_bench.py 4096 8 nosweep > $OUT/pmcIB.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmcIB/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k[0], k[1], "%.4g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
