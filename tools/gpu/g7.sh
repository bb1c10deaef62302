cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_wb_gpu.py tests/test_backend_gpu.py -m gpu -q -s > gpurun_out/g7_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g7_tests.log
grep -E "^E  |^FAILED|passed|failed|worst|clamped|rel dev" gpurun_out/g7_tests.log | head -40
