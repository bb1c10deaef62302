cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/g4_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g4_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/g4_tests.log | head -20
for k in 3 4 6; do
  echo "pg_steps=$k: $(python bench.py --warm-start-steps $k --no-cpu-baseline --no-extras --steps 300 --warmup 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us, ws iters", round(d["working_set_iters_mean"],4), "unsolved", d["unsolved_problems"], "in_order", round(d["in_order"]["ms_per_step"]*1e3,2), d["parity_spot_check"]["worst_rel"])')"
done 2>&1 | tee gpurun_out/g4_pg.txt
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("driver flags:", round(d["ms_per_step"]*1e3,2), "kernel", round(d["roofline"]["kernel_ms_avg"]*1e3,2), "steady", round(d["steady_state"]["ms_per_step"]*1e3,2), "in_order", round(d["in_order"]["ms_per_step"]*1e3,2))'
