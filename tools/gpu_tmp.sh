cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dev
T="tests/test_friction_cone.py tests/test_contact_wrench_cone.py tests/test_golden_ref.py tests/test_gpu_parity.py tests/test_newton_iteration.py tests/test_shapes.py tests/test_acceleration_limits.py tests/test_stage_dump.py tests/test_determinism.py tests/test_random_grids.py"
for split in 0 1; do RTOC_PARITY_PINS=0 RTOC_CONDENSE_SPLIT=$split timeout 600 python -m pytest $T -m gpu -q -x 2>&1 | tail -12 > gpurun_out/dev/pytest_s$split.log; grep -n "passed\|failed\|FAILED" gpurun_out/dev/pytest_s$split.log; done
for i in 1 2 3; do for split in 0 1; do echo "split=$split"; RTOC_CONDENSE_SPLIT=$split timeout 200 python tools/sqp_bench.py 4096 2>&1 | tail -1; done; done
python tools/icub_condense_bench.py | grep icub
