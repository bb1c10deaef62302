cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 0 2 3 4 5 6 8; do
  echo "pg_steps=$k: $(python bench.py --warm-start-steps $k --no-cpu-baseline --no-extras --steps 300 --warmup 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us, ws iters", round(d["working_set_iters_mean"],4), "unsolved", d["unsolved_problems"], "in_order", round(d["in_order"]["ms_per_step"]*1e3,2))')"
done 2>&1 | tee gpurun_out/g3_pg.txt
