#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_i}
mkdir -p $OUT
for pg in 2 3 4 5 6; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged --warm-start-steps $pg > $OUT/bench_pg$pg.json 2>> $OUT/err.txt
  python - <<PY
import json
d=json.load(open("$OUT/bench_pg$pg.json"))
print("pg=$pg us/step %.2f"%(d["ms_per_step"]*1e3), "kernel_ms %.4f"%d["roofline"]["kernel_ms_per_launch"], "steady %.3f"%(d["steady_state"]["ms_per_step"]*1e3), "in_order %.2f"%(d["in_order"]["ms_per_step"]*1e3), d["working_set_iters_mean"], d["unsolved_problems"])
PY
done
for ns in 6000 9000 11000 13700 17000; do
  echo "stagger $ns:" $(ALORE_NMPC_STAGGER_NS=$ns python tools/launch_overhead.py 20 2>/dev/null | grep eager | tail -1)
done
