cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pinned or stress or every_lane or golden or wide" > gpurun_out/g10_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g10_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/g10_tests.log | head
python bench.py --no-cpu-baseline --no-extras --no-converged --steps 300 --warmup 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("300 steps:", round(d["ms_per_step"]*1e3,3), "us; in_order", round(d["in_order"]["ms_per_step"]*1e3,2), d["parity_spot_check"]["worst_rel"])'
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-converged 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("driver flags:", round(d["ms_per_step"]*1e3,2), "kernel", round(d["roofline"]["kernel_ms_avg"]*1e3,2), "steady", round(d["steady_state"]["ms_per_step"]*1e3,2))'
