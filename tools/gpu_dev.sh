#!/bin/bash
# Development run on the GPU box: the whole GPU suite on both condensation pipelines (RTOC_CONDENSE_SPLIT=1 two kernels,
# =0 the fused kernel), SQP phase timings and cycle stamps of both.   gpurun -- bash tools/gpu_dev.sh [quick]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/dev
mkdir -p $OUT
cd $R
for split in 1 0; do
  if [ "${1:-}" != "quick" ]; then
    RTOC_PARITY_PINS=${PINS:-0} RTOC_CONDENSE_SPLIT=$split timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest_split$split.log
    echo "== split=$split"; tail -4 $OUT/pytest_split$split.log
  fi
  RTOC_CONDENSE_SPLIT=$split timeout 200 python tools/sqp_bench.py 4096 2>&1 | tail -1 | tee $OUT/sqp_split$split.log
  if [ -f $R/robotoc_amd/librtoc_hip_prof.so ]; then
    RTOC_HIP_LIB=$R/robotoc_amd/librtoc_hip_prof.so RTOC_CONDENSE_SPLIT=$split timeout 200 python tools/condense_profile.py 2>&1 | tail -3 | tee $OUT/prof_split$split.log
  fi
done
