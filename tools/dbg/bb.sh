#!/bin/bash
python tools/dbg/bb_probe.py
for n in 3 4 5 6 7 8 10; do
  export ALORE_NMPC_PG_BB=$n
  python tools/dbg/bb_probe.py 2>&1 | tail -1
  echo "  BB=$n: $(python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us")' 2>&1)"
done
unset ALORE_NMPC_PG_BB
echo "  fixed-step 8: $(python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us")' 2>&1)"
