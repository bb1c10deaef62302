cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_wb_gpu.py tests/test_gpu_parity.py -m gpu -q -s -k "contact_rows or pinned or penalty" > gpurun_out/g9_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g9_tests.log
grep -E "^E  |^FAILED|passed|failed|contact rows|stance-foot" gpurun_out/g9_tests.log | head -30
