cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_ltv_mpc.py -m gpu -q -k "stress or groups_of or non_finite" > gpurun_out/g2_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g2_tests.log
grep -E "^E  |^FAILED|passed|failed|QPs" gpurun_out/g2_tests.log | head -40
