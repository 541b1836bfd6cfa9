cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/rv_bench.py 4096 trot > gpurun_out/a1.log 2>&1
RTOC_EXPERIMENT_FORCE_STO_KERNEL=1 python tools/rv_bench.py 4096 trot > gpurun_out/b1.log 2>&1
python tools/rv_bench.py 4096 trot > gpurun_out/a2.log 2>&1
RTOC_EXPERIMENT_FORCE_STO_KERNEL=1 python tools/rv_bench.py 4096 trot > gpurun_out/b2.log 2>&1
grep "register\|worst" gpurun_out/a1.log gpurun_out/b1.log gpurun_out/a2.log gpurun_out/b2.log
