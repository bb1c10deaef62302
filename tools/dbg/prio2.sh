#!/bin/bash
for B in 1024 2048 8192 16384; do for v in libalore_nmpc_base v_prioB; do
  export ALORE_NMPC_LIB=$PWD/ab/$v.so
  echo "B=$B $v: $(python bench.py --batch $B --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us")' 2>&1)"
done; done
