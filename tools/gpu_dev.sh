#!/bin/bash
# Development run on the GPU box: the condensation-related GPU tests on both pipelines (RTOC_CONDENSE_SPLIT=1 two kernels,
# =0 the fused kernel), the new determinism tests, SQP phase timings of both.   gpurun -- bash tools/gpu_dev.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/dev
mkdir -p $OUT
cd $R
T="tests/test_friction_cone.py tests/test_contact_wrench_cone.py tests/test_golden_ref.py tests/test_gpu_parity.py tests/test_newton_iteration.py tests/test_contact_closed_loop.py tests/test_shapes.py tests/test_contact_constraints.py tests/test_sto_closed_loop.py tests/test_cpp_solver.py tests/test_stage_dump.py"
for split in 0 1; do
  RTOC_PARITY_PINS=0 RTOC_CONDENSE_SPLIT=$split timeout 600 python -m pytest $T -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest_split$split.log
  echo "== split=$split"; tail -4 $OUT/pytest_split$split.log
  RTOC_CONDENSE_SPLIT=$split timeout 200 python tools/sqp_bench.py 4096 2>&1 | tail -1 | tee $OUT/sqp_split$split.log
  RTOC_HIP_LIB=$R/robotoc_amd/librtoc_hip_prof.so RTOC_CONDENSE_SPLIT=$split timeout 200 python tools/condense_profile.py 2>&1 | tail -3 | tee $OUT/prof_split$split.log
done
timeout 900 python -m pytest tests/test_determinism.py -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_determinism.log
