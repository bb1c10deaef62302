#!/bin/bash
# wavefronts per workgroup: 1 vs 4
for B in 2048 4096 8192 32768; do for w in 1 4; do
  export ALORE_NMPC_WPB=$w
  echo "B=$B wpb=$w: $(python bench.py --batch $B --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", d["config"]["threads_per_block"], d["config"]["lds_bytes_per_block"])' 2>&1)"
done; done
