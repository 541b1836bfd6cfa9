#!/bin/bash
# One GPU-box round (round 2): parity tests, bench, rocprofv3 kernel stats of the WHOLE bench command (headline
# sweep + SQP iteration + the other configs: every kernel DESIGN.md quotes), HBM traffic (FETCH_SIZE / WRITE_SIZE in
# their own passes, as MI355X_MICROARCH.md prescribes) and one SQ counter pass.  Env: SKIP_TESTS, SKIP_PROF, SKIP_PMC.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $OUT/host.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 ${PYTEST_ARGS:-} 2>&1 | tail -40 > $OUT/pytest_gpu.log
fi
timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  export TMPDIR=/tmp
  cd /tmp
  B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- $B > $OUT/prof_stats.log 2>&1
  if [ "${SKIP_PMC:-0}" != "1" ]; then
    B3="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
    timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- $B3 > $OUT/prof_fetch.log 2>&1
    timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- $B3 > $OUT/prof_write.log 2>&1
    timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/prof_sq1 -o sq1 -- $B3 > $OUT/prof_sq1.log 2>&1
    timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/prof_sq2 -o sq2 -- $B3 > $OUT/prof_sq2.log 2>&1
  fi
  cd $R
fi
cat $OUT/pytest_gpu.log 2>/dev/null | tail -12
head -c 3000 $OUT/bench.json; echo; tail -3 $OUT/bench.err
