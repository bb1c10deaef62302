#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_c}
mkdir -p $OUT
for pv in 0 1 0 1; do
  ALORE_NMPC_PERSIST=$pv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged > $OUT/bench_p${pv}.json 2>> $OUT/err.txt
  python - <<PY
import json
d=json.load(open("$OUT/bench_p${pv}.json"))
print("persist=$pv", "us/step %.2f"%(d["ms_per_step"]*1e3), "frac %.3f"%d["roofline"]["frac"], "kernel_ms %.4f"%d["roofline"]["kernel_ms_per_launch"], "steady %.3f"%(d["steady_state"]["ms_per_step"]*1e3), d["host_clock_breakdown_us"], d["parity_spot_check"]["ok"], d["unsolved_problems"])
PY
done
ALORE_NMPC_PERSIST=1 python tools/launch_overhead.py 20 > $OUT/launch_overhead_p1.txt 2>> $OUT/err.txt
ALORE_NMPC_PERSIST=0 python tools/launch_overhead.py 20 > $OUT/launch_overhead_p0.txt 2>> $OUT/err.txt
cat $OUT/launch_overhead_p1.txt $OUT/launch_overhead_p0.txt
ALORE_NMPC_PERSIST=1 ALORE_NMPC_TRACE=$OUT/tr20 python tools/trace_grid.py 20 2 > $OUT/timeline_20_p1.txt 2>> $OUT/err.txt
ALORE_NMPC_PERSIST=1 ALORE_NMPC_TRACE=$OUT/tr200 python tools/trace_grid.py 200 1 > $OUT/timeline_200_p1.txt 2>> $OUT/err.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned or stress or mask" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
grep -E "duration|SIMD-time|before the first|gap|lifetime \(" $OUT/timeline_20_p1.txt $OUT/timeline_200_p1.txt
