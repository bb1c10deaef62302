#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_j}
mkdir -p $OUT
for pg in 3 4 5 6 4; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged --warm-start-steps $pg > $OUT/bench_pg$pg.json 2>> $OUT/err.txt
  python - <<PY
import json
d=json.load(open("$OUT/bench_pg$pg.json"))
print("pg=$pg us/step %.2f"%(d["ms_per_step"]*1e3), "kernel_ms %.4f"%d["roofline"]["kernel_ms_per_launch"], "steady %.3f"%(d["steady_state"]["ms_per_step"]*1e3), "in_order %.2f"%(d["in_order"]["ms_per_step"]*1e3), d["working_set_iters_mean"], d["unsolved_problems"], d["parity_spot_check"]["ok"])
PY
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned or stress or mask or one_tick" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
