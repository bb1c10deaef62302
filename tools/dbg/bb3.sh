#!/bin/bash
# adaptive BB (positive steps: stop when settled, cap 2x) vs fixed count (negative)
for n in 4 5 6 7; do
  echo "steps=$n: B=4096 $(python bench.py --warm-start-steps $n --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", d["working_set_iters_mean"])' 2>&1)  B=32768 $(python bench.py --batch 32768 --warm-start-steps $n --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,1), "us", d["working_set_iters_mean"])' 2>&1)  N=50 $(python bench.py --horizon 50 --warm-start-steps $n --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,1), "us", d["working_set_iters_mean"])' 2>&1)"
done
