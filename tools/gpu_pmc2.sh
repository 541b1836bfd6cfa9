#!/bin/bash
# PMC passes (each --pmc set in its own run; no tracing domains mixed in) over tools/pmc_driver.py, plus a probe of
# how the CPU baseline scales with the thread count on this host (cgroup quota / affinity recorded).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{ echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; python -c "import os; print('affinity:', len(os.sched_getaffinity(0)))"; } > $OUT/host_cpu.txt
python - > $OUT/cpu_scaling.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from robotoc_amd import problems as pr
from oracle import oracle as orc
dims, grids, _ = pr.config_anymal_trot()
L = orc.layout(dims)
k = pr.make_kkt_batch_unique(L, grids, 512, seed=99); dx0 = pr.make_dx0_unique(L, 512, seed=99)
for nt in (1, 8, 16, 32, 64, 128, 256):
    orc.bench_sweep(L, grids, k, dx0, 1, nt)
    best = 0
    for _ in range(3):
        r = orc.bench_sweep(L, grids, k, dx0, max(1, nt // 8), nt)
        best = max(best, r["sweeps"] / r["seconds"])
    print(nt, "threads:", round(best, 1), "sweeps/s")
PY
export TMPDIR=/tmp
cd /tmp
D="python $R/tools/pmc_driver.py 3"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- $D > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- $D > $OUT/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/prof_sq1 -o sq1 -- $D > $OUT/prof_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/prof_sq2 -o sq2 -- $D > $OUT/prof_sq2.log 2>&1
cd $R
cat $OUT/host_cpu.txt $OUT/cpu_scaling.txt
tail -n 2 $OUT/prof_fetch.log; tail -n 2 $OUT/prof_sq2.log
