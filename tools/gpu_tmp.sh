cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dev
RTOC_PARITY_PINS=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/dev/pytest_all.log
grep -n "passed\|failed" gpurun_out/dev/pytest_all.log
RTOC_HIP_LIB=$GRAFT_REPO_ROOT/robotoc_amd/librtoc_hip_prof.so python tools/expand_profile.py | tail -2
