cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_backward_register_wide.py -x -q 2>&1 | tail -3 > gpurun_out/rw_test.log
python tools/rv_bench.py 1024 icub32 > gpurun_out/rw_bench32.log 2>&1
RTOC_HIP_LIB=robotoc_amd/librtoc_hip_prof.so python tools/phase_profile_rv.py 1024 icub32 > gpurun_out/rw_prof32.log 2>&1
cat gpurun_out/rw_test.log gpurun_out/rw_bench32.log gpurun_out/rw_prof32.log
