cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/pytest_gpu_full.log
grep -n "FAILED\|passed\|failed\|ERROR" gpurun_out/pytest_gpu_full.log
