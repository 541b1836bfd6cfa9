#!/bin/bash
# The round's closing GPU run: full -m gpu suite, bench line, rocprofv3 kernel statistics of the whole bench command, the
# counter passes (tools/gpu_pmc2.sh: FETCH / WRITE / two SQ sets over tools/pmc_driver.py), kernel statistics of the closed-loop
# iteration, and the summaries written ON THE BOX into gpurun_out/summary (the raw traces exceed what gpurun copies back).
# Afterwards, locally:  cp gpurun_out/summary/* profiles/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r02}
mkdir -p $OUT
cd $R
SKIP_PMC=1 bash tools/gpu_round2.sh > $OUT/round.log 2>&1
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
bash tools/gpu_pmc2.sh > $OUT/pmc2.log 2>&1
bash tools/gpu_closed_loop_prof.sh > $OUT/closed_loop.log 2>&1
mkdir -p $OUT/summary
RTOC_PROFILE_OUT=$OUT/summary python tools/summarize_profiles.py $TAG "closing run of the round" > $OUT/summarize.log 2>&1
cp $OUT/closed_loop_kernel_stats.txt $OUT/summary/${TAG}_closed_loop_kernel_stats.txt
cp $OUT/host_cpu.txt $OUT/summary/${TAG}_host_cpu.txt 2>/dev/null
cat $OUT/cpu_scaling.txt >> $OUT/summary/${TAG}_host_cpu.txt 2>/dev/null
cp $OUT/pytest_gpu.log $OUT/summary/${TAG}_pytest_gpu.log
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq1 $OUT/prof_sq2
ls -la $OUT/summary
tail -2 $OUT/summarize.log
head -c 600 $OUT/bench.json; echo
