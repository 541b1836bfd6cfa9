cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dev
RTOC_PARITY_PINS=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/dev/pytest_all.log
grep -n "passed\|failed\|FAILED" gpurun_out/dev/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
