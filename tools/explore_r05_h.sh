#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_h}
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned or stress or mask or one_tick" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged > $OUT/bench_$i.json 2>> $OUT/err.txt
  python - <<PY
import json
d=json.load(open("$OUT/bench_$i.json"))
print("us/step %.2f"%(d["ms_per_step"]*1e3), "frac %.3f"%d["roofline"]["frac"], "kernel_ms %.4f"%d["roofline"]["kernel_ms_per_launch"], "steady %.3f"%(d["steady_state"]["ms_per_step"]*1e3), "in_order %.2f"%(d["in_order"]["ms_per_step"]*1e3), d["host_clock_breakdown_us"], d["parity_spot_check"]["ok"], d["unsolved_problems"], d["working_set_iters_mean"], d["config"]["hip_graph"])
PY
done
ALORE_NMPC_TRACE=$OUT/tr200 python tools/trace_grid.py 200 1 > $OUT/timeline_200.txt 2>> $OUT/err.txt
ALORE_NMPC_TRACE=$OUT/tr20 python tools/trace_grid.py 20 2 > $OUT/timeline_20.txt 2>> $OUT/err.txt
grep -E "duration|lifetime \(|compute|loads \(|gap|SIMD-time|before the" $OUT/timeline_200.txt $OUT/timeline_20.txt
