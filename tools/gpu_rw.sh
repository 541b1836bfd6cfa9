cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for s in 0 127 254 0 127 64 381; do
RTOC_RV_STAGGER=$s python tools/rv_bench.py 4096 trot 2>&1 | grep "register\|worst" | sed "s/^/stagger $s: /"
done > gpurun_out/stagger.log
cat gpurun_out/stagger.log
