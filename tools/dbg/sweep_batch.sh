#!/bin/bash
# batch-size sweep (diagnostic): how launch time grows with waves per SIMD
for L in 32 16; do
  for B in 64 512 1024 2048 4096 8192 16384; do
    echo "L=$L B=$B: $(python bench.py --batch $B --lanes $L --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", d["value"])' 2>&1)"
  done
done
