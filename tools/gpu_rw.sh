cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_backward_register_wide.py tests/test_backward_register.py -x -q 2>&1 | grep -v amdgpu.ids | tail -1
for c in icub32 icub35; do echo "== $c (Fxx structure asserted by the caller)"; RTOC_FXX=2 timeout 200 python tools/rv_bench.py 1024 $c 2>&1 | grep "backward\|worst"; done
