#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_o}
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
for sh in 0 1 0 1; do
  ALORE_NMPC_XCD_DEBUG=1 ALORE_NMPC_XCD_SHARES=$sh python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged > $OUT/b$sh.json 2> $OUT/b$sh.err
  python - <<PY
import json
d=json.load(open("$OUT/b$sh.json")); r=d["roofline"]
print("shares=$sh us/step %.2f"%(d["ms_per_step"]*1e3), "frac %.3f"%r["frac"], "kernel_ms %.4f"%r["kernel_ms_per_launch"], "steady %.3f frac %.3f"%(d["steady_state"]["ms_per_step"]*1e3, d["steady_state"]["hbm_frac"]), d["parity_spot_check"]["ok"], d["parity_stress_check"]["ok"])
PY
  grep "xcd shares" $OUT/b$sh.err | tail -2
done
ALORE_NMPC_XCD_DEBUG=1 ALORE_NMPC_TRACE=$OUT/tr python tools/trace_grid.py 200 3 2>$OUT/tr.err | grep -E "^====|duration|XCD [0-7]|after the last" 
grep "xcd shares" $OUT/tr.err
