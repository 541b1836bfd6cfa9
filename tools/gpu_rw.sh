cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_backward_register_wide.py -x -q -k "35" 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/rw_test.log
timeout 200 python tools/rv_bench.py 1024 icub35 > gpurun_out/rw_bench35.log 2>&1
RTOC_HIP_LIB=robotoc_amd/librtoc_hip_prof.so timeout 200 python tools/phase_profile_rv.py 1024 icub35 > gpurun_out/rw_prof35.log 2>&1
cat gpurun_out/rw_test.log; grep "backward\|worst" gpurun_out/rw_bench35.log; cat gpurun_out/rw_prof35.log | grep -v amdgpu | head -8
