cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_backward_register_wide.py tests/test_backward_register.py tests/test_bench_launcher.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/rw_test.log
python tools/rv_bench.py 4096 trot > gpurun_out/rv_bench_check.log 2>&1
RTOC_FXX=2 python tools/rv_bench.py 4096 trot > gpurun_out/rv_bench_assert.log 2>&1
python tools/rv_bench.py 4096 trot >> gpurun_out/rv_bench_check.log 2>&1
RTOC_FXX=2 python tools/rv_bench.py 4096 trot >> gpurun_out/rv_bench_assert.log 2>&1
cat gpurun_out/rw_test.log; grep register gpurun_out/rv_bench_check.log; echo ---; grep register gpurun_out/rv_bench_assert.log
