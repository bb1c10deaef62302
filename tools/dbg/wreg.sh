#!/bin/bash
for B in 4096 8192 32768; do for w in 0 1; do
  export ALORE_NMPC_WREG=$w
  echo "B=$B wreg=$w: $(python bench.py --batch $B --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", d["config"]["lds_bytes_per_block"])' 2>&1)"
done; done
