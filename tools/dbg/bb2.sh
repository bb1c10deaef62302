#!/bin/bash
# BB predictor steps: N = 50 and large batch
for n in 5 6 7 8 10 12 16; do
  export ALORE_NMPC_PG_BB=$n
  echo "BB=$n N=50: $(python bench.py --horizon 50 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,1), "us", d["working_set_iters_mean"])' 2>&1)   B=32768: $(python bench.py --batch 32768 --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,1), "us", d["working_set_iters_mean"])' 2>&1)"
done
