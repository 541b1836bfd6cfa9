#!/bin/bash
# kernel-level breakdown of the scan path (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in ${CFGS:-anymal icub32}; do
  python $R/tools/scan_latency.py $cfg 1 50 > $OUT/scan_lat_$cfg.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_prof_$cfg -o p -- python $R/tools/scan_latency.py $cfg 1 20 > $OUT/scan_prof_$cfg.log 2>&1
  f=$(find $OUT/scan_prof_$cfg -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; cat $OUT/scan_lat_$cfg.log; cut -d, -f1-6 $f | sed 's/rtoc:://; s/(rtoc::[A-Za-z]*)//' | head -8
done
