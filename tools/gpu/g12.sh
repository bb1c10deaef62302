cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/micro/pk_mov_check
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/g12_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g12_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/g12_tests.log | head
python bench.py --no-cpu-baseline --no-extras --no-converged --steps 300 --warmup 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("300 steps:", round(d["ms_per_step"]*1e3,3), "us; in_order", round(d["in_order"]["ms_per_step"]*1e3,2), d["parity_spot_check"]["worst_rel"])'
