#!/bin/bash
# Register-resident backward kernel on the GPU box: timing beside the role-split kernel, per-kernel durations (rocprofv3), cycle stamps.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rv
mkdir -p $OUT
cd $R
if [ "${1:-}" = "test" ]; then timeout 600 python -m pytest tests/test_backward_register.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.log; fi
timeout 300 python tools/rv_bench.py 4096 2>&1 | tail -12 | tee $OUT/bench.log
( export TMPDIR=/tmp; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o rv -- python $R/tools/rv_bench.py 4096 > $OUT/rocprof.log 2>&1 )
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
{ echo "kernel, calls, total ns, avg ns, %, min, max, stddev"; grep -i "riccati" $F; } | cut -c1-220 | tee $OUT/kernel_stats.txt
rm -rf $OUT/prof
RTOC_HIP_LIB=$R/robotoc_amd/librtoc_hip_prof.so timeout 200 python tools/phase_profile_rv.py 4096 2>&1 | tail -14 | tee $OUT/prof.log
