#!/bin/bash
for B in 2048 4096 8192 32768; do for L in 32 64; do
  echo "B=$B L=$L: $(python bench.py --batch $B --lanes $L --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,1), "us", d["config"]["lds_bytes_per_block"])' 2>&1)"
done; done
