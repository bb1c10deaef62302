cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_host_layer.py tests/test_gpu_parity.py tests/test_closed_loop.py tests/test_cpp_host.py tests/test_acado_compat.py tests/test_config4_pipeline.py -m gpu -q > gpurun_out/g16_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g16_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/g16_tests.log | head -30
