#!/bin/bash
# timing-only experiments on the register kernel (debug builds: parts of the memory traffic switched off by a mask; nt = non-temporal)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for lib in dbg dbgnt; do
for m in ${MASKS:-0 1 31}; do
  echo "$lib mask $m: $(RTOC_RV_DEBUG=$m RTOC_HIP_LIB=$R/robotoc_amd/librtoc_hip_$lib.so timeout 120 python tools/rv_bench.py 4096 2>&1 | grep '^register' | tail -1)"
done
done
