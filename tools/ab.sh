#!/bin/bash
# A/B timing of two builds of libalore_nmpc.so on the same box: ab/libalore_nmpc_base.so vs the in-tree one
B=${1:-4096}
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export ALORE_NMPC_LIB=$PWD/ab/libalore_nmpc_base.so; else unset ALORE_NMPC_LIB; fi
    echo "$v B=$B: $(python bench.py --batch $B --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us")' 2>&1)"
  done
done
