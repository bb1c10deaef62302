cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/g11_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g11_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/g11_tests.log | head
