#!/bin/bash
for ws in 8 12 16 24 32; do
  echo "N=50 L=64 pg=$ws: $(python bench.py --horizon 50 --warm-start-steps $ws --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,1), "us", d["working_set_iters_mean"])' 2>&1)"
done
