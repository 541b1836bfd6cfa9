#!/bin/bash
# The round's closing GPU run: full -m gpu suite, bench line, rocprofv3 kernel statistics of the whole bench command, the
# counter passes (tools/gpu_pmc2.sh: FETCH / WRITE / two SQ sets over tools/pmc_driver.py), kernel statistics of the closed-loop
# iteration, and the summaries written ON THE BOX into gpurun_out/summary (the raw traces exceed what gpurun copies back).
# Afterwards, locally:  cp gpurun_out/summary/* profiles/
# After a kernel change that moves a summation order: record the regression pins again first --
#   gpurun -- 'RTOC_PARITY_PINS=0 python -m pytest tests -q -m gpu'  &&  python tools/make_parity_pins.py   (tests/golden/parity_pins.json)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r06}
mkdir -p $OUT
cd $R
SKIP_PMC=1 bash tools/gpu_round2.sh > $OUT/round.log 2>&1
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
bash tools/gpu_pmc2.sh > $OUT/pmc2.log 2>&1
bash tools/gpu_closed_loop_prof.sh > $OUT/closed_loop.log 2>&1
mkdir -p $OUT/summary
bash tools/gpu_pmc_lin_flops.sh $TAG 1024 > $OUT/lin_flops.log 2>&1
# kernel statistics of the single-instance scan on the grid with switching-time optimisation (ANYmal jump_sto), and its phase stamps
( export TMPDIR=/tmp; cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sto -o sto -- python $R/tools/scan_sto_latency.py > $OUT/scan_sto_latency.txt 2>&1 )
{ cat $OUT/scan_sto_latency.txt | grep "bwd/fwd"; echo; echo "kernel, calls, total ns, avg ns, %, min, max, stddev"; grep -i "sto\|riccati_backward\|scan_" $(find $OUT/prof_sto -name "*kernel_stats.csv" | head -1); } > $OUT/summary/${TAG}_scan_sto_kernel_stats.txt 2>/dev/null
if [ -f $R/robotoc_amd/librtoc_hip_prof.so ]; then RTOC_HIP_LIB=$R/robotoc_amd/librtoc_hip_prof.so timeout 100 python tools/phase_profile_sto.py >> $OUT/summary/${TAG}_scan_sto_kernel_stats.txt 2>&1; fi
rm -rf $OUT/prof_sto
timeout 200 python tools/dvfs_probe.py > $OUT/summary/${TAG}_dvfs_probe.txt 2>&1
# the register-chained condensation kernel beside the role-split one: with rows and cones, without, per-kernel durations, cycle stamps
{ echo "== 72 joint-limit rows + 4 friction cones =="; bash tools/gpu_cond.sh prof 2>&1 | grep -v amdgpu.ids; echo "== no rows =="; bash tools/gpu_cond.sh x norows 2>&1 | grep -v amdgpu.ids;
  if [ -f $R/robotoc_amd/librtoc_hip_prof.so ]; then echo "== cycle stamps of one work item (two waves per SIMD) =="; RTOC_HIP_LIB=$R/robotoc_amd/librtoc_hip_prof.so timeout 100 python tools/phase_profile_cond.py 4096 2>&1 | grep -v amdgpu.ids; fi; } > $OUT/summary/${TAG}_condense_register.txt 2>&1
timeout 200 python tools/icub_bwd_bench.py > $OUT/summary/${TAG}_icub_backward.txt 2>&1
# cycle stamps of the register-wide kernels (both waves of the nv = 35 one), Fxx structure asserted
if [ -f $R/robotoc_amd/librtoc_hip_prof.so ]; then for c in icub35 icub32; do echo "== $c"; RTOC_FXX=2 RTOC_HIP_LIB=$R/robotoc_amd/librtoc_hip_prof.so timeout 200 python tools/phase_profile_rv.py 1024 $c 2>&1 | grep -v amdgpu | head -12; done > $OUT/summary/${TAG}_register_wide_phase_stamps.txt; fi
# run-to-run determinism of the headline sweep, records compared bit for bit (instance / stage / field of anything that differs)
timeout 150 python tools/determinism_probe.py 40 > $OUT/summary/${TAG}_determinism.txt 2>&1
RTOC_PROFILE_OUT=$OUT/summary python tools/summarize_profiles.py $TAG "closing run of the round" > $OUT/summarize.log 2>&1
cp $OUT/closed_loop_kernel_stats.txt $OUT/summary/${TAG}_closed_loop_kernel_stats.txt
cp $OUT/host_cpu.txt $OUT/summary/${TAG}_host_cpu.txt 2>/dev/null
cat $OUT/cpu_scaling.txt >> $OUT/summary/${TAG}_host_cpu.txt 2>/dev/null
cp $OUT/pytest_gpu.log $OUT/summary/${TAG}_pytest_gpu.log
# the bench line once more, now that the counter passes of THESE kernel sources exist: the driver's own run at round end reads
# the committed profiles/<tag>_traffic.json / _linearize_flops.json, this is the same line taken on this box
cp $OUT/summary/${TAG}_traffic.json $OUT/summary/${TAG}_linearize_flops.json $R/profiles/ 2>/dev/null
timeout 600 python bench.py > $OUT/summary/${TAG}_bench.json 2> $OUT/bench2.err   # the compact line the driver parses ...
cp $OUT/bench_detail.json $OUT/summary/${TAG}_bench_detail.json 2>/dev/null             # ... and the full record behind it
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq1 $OUT/prof_sq2
ls -la $OUT/summary
tail -2 $OUT/summarize.log
head -c 600 $OUT/summary/${TAG}_bench.json; echo
