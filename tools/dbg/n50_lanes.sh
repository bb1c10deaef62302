#!/bin/bash
for L in 64 32 16; do
  for ws in -1 0; do
  echo "N=50 L=$L pg=$ws: $(python bench.py --horizon 50 --lanes $L --warm-start-steps $ws --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,1), "us", d["working_set_iters_mean"], d["config"]["lds_bytes_per_block"])' 2>&1)"
  done
done
