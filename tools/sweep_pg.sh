#!/bin/bash
# B = 4096 launch time against the number of working-set prediction steps (alore_nmpc_config.warm_start_steps)
for k in 0 4 6 8 10 12 16; do
  echo "pg_steps=$k: $(python bench.py --warm-start-steps $k --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us, ws iters", round(d["working_set_iters_mean"],3), "unsolved", d["unsolved_problems"])')"
done
