#!/bin/bash
# prediction-step sweep at the bench batch
for B in 4096 32768; do
for pg in 4 5 6 7 8 10; do
  echo "pg=$pg B=$B: $(python bench.py --batch $B --warm-start-steps $pg --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", d["working_set_iters_mean"])' 2>&1)"
done
done
