#!/bin/bash
# One GPU-box round: parity tests, bench, rocprofv3 kernel stats, PMC passes (HBM bytes).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu.log
fi
timeout 600 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  export TMPDIR=/tmp
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sqp --no-configs ${BENCH_ARGS:-} > $OUT/prof_stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sqp --no-configs ${BENCH_ARGS:-} > $OUT/prof_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sqp --no-configs ${BENCH_ARGS:-} > $OUT/prof_write.log 2>&1
  cd $R
  find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -name "*.csv" | head -20 > $OUT/prof_files.txt
fi
cat $OUT/pytest_gpu.log 2>/dev/null | tail -5
cat $OUT/bench.json; tail -3 $OUT/bench.err
