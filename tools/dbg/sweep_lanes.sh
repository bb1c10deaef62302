#!/bin/bash
# lanes-per-problem sweep (diagnostic): python bench.py for B in {4096, 32768}, L in {4,8,16,32}
for B in 4096 32768; do
  for L in 4 8 16 32; do
    echo "B=$B L=$L: $(python bench.py --batch $B --lanes $L --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"]*1e3, "us", d["value"])' 2>&1)"
  done
done
