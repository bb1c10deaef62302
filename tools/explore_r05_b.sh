#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_b}
mkdir -p $OUT
tools/micro/dispatch_gap > $OUT/dispatch_gap.txt 2>&1
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged > $OUT/bench_driver_$i.json 2>> $OUT/err.txt
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged --graph always > $OUT/bench_driver_graph.json 2>> $OUT/err.txt
ALORE_NMPC_TRACE=$OUT/tr20 python tools/trace_grid.py 20 2 > $OUT/timeline_20.txt 2>> $OUT/err.txt
ALORE_NMPC_TRACE=$OUT/tr200 python tools/trace_grid.py 200 1 > $OUT/timeline_200.txt 2>> $OUT/err.txt
python -m pytest tests/test_gpu_parity.py tests/test_host_layer.py -x -q -m gpu -k "mask or idle or pinned" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
cat $OUT/dispatch_gap.txt
python - <<PY
import json
for f in ("bench_driver_1","bench_driver_2","bench_driver_graph"):
    d=json.load(open("$OUT/%s.json"%f))
    print(f, d["ms_per_step"]*1e3, d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["config"]["hip_graph"], d.get("steady_state",{}).get("ms_per_step"), d.get("in_order",{}).get("ms_per_step"))
PY
