#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_m}
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_masked_regions.py -x -q -m gpu > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-converged > $OUT/bench.json 2>> $OUT/err.txt
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("us/step %.2f"%(d["ms_per_step"]*1e3), "steady %.3f"%(d["steady_state"]["ms_per_step"]*1e3), "in_order %.2f us, kernel %.2f"%(d["in_order"]["ms_per_step"]*1e3, d["in_order"]["kernel_ms_avg"]*1e3))
print({k:(v["p50_ms"],v["p99_ms"]) for k,v in d["latency"].items() if isinstance(v,dict)})
print(d["extras"].get("warm_tick"), d["extras"].get("device_closed_loop"))
PY
ALORE_NMPC_STAMPS=1 python tools/ab_block.py --time --lanes 16 --B 4096 2>&1 | grep -A12 "stamps\]" | head -30
ALORE_NMPC_STAMPS=1 python tools/ab_block.py --lanes 32 --B 1 2>&1 | grep -A12 "stamps\]" | head -14
