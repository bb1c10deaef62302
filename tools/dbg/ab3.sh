#!/bin/bash
# A/B/C timing: ab/libalore_nmpc_base.so, ab/libalore_nmpc_w3.so, in-tree
for B in 4096 8192 32768; do
for rep in 1 2; do
  for v in base new w3; do
    if false; then unset ALORE_NMPC_LIB; else export ALORE_NMPC_LIB=$PWD/ab/libalore_nmpc_$v.so; fi
    echo "$v B=$B: $(python bench.py --batch $B --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", d["config"]["lds_bytes_per_block"])' 2>&1)"
  done
done
done
