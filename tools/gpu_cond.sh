#!/bin/bash
# Register-chained condensation kernel on the GPU box: parity tests, timing beside the role-split kernel, per-kernel durations.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/cond
mkdir -p $OUT
cd $R
timeout 300 python tools/cond_bench.py 4096 ${2:-} 2>&1 | tail -14 | tee $OUT/bench.log
if [ "${1:-}" = "test" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "condens or golden or friction or acceleration or pdipm or random_event or shapes or wrench or newton or closed_loop" 2>&1 | grep -v "^  test_\|^$" | tail -25 | tee $OUT/pytest.log; fi
if [ "${1:-}" = "prof" ]; then
( export TMPDIR=/tmp; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c -- python $R/tools/cond_bench.py 4096 ${2:-} > $OUT/rocprof.log 2>&1 )
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
{ echo "kernel, calls, total ns, avg ns, %, min, max, stddev"; grep -i "condense\|mjtjinv" $F; } | cut -c1-220 | tee $OUT/kernel_stats.txt
rm -rf $OUT/prof
fi
