#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_n}
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python - <<PY
import json
for f in ("bench_default","bench_driver"):
    d=json.load(open("$OUT/%s.json"%f))
    print(f, "us/step %.2f"%(d["ms_per_step"]*1e3), "value %.3e"%d["value"], "frac %.3f"%d["roofline"]["frac"], d.get("error"), d["parity_spot_check"]["ok"], d["parity_stress_check"]["ok"], d["host_clock_breakdown_us"])
    ex=d.get("extras",{})
    print("  extras:", [k for k in ex], [k for k,v in ex.items() if isinstance(v,dict) and "error" in v])
    print("  8192:", ex.get("backend_to_nmpc_pipeline_8192"))
PY
