#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_l}
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python bench.py --steps 20 --warmup 5 --headline converged_all_gather --no-extras --no-cpu-baseline > $OUT/bench_conv.json 2> $OUT/bench_conv.err
python - <<PY
import json
d=json.load(open("$OUT/bench_driver.json"))
print("us/step %.2f"%(d["ms_per_step"]*1e3), "value %.3e"%d["value"], "frac %.3f"%d["roofline"]["frac"], "kernel_ms %.4f"%d["roofline"]["kernel_ms_per_launch"], "steady %.3f"%(d["steady_state"]["ms_per_step"]*1e3), d["steady_state"]["hbm_frac"], "in_order %.2f"%(d["in_order"]["ms_per_step"]*1e3))
print(d["parity_spot_check"]); print(d["parity_stress_check"]); print(d.get("error"))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!="what"}) for k,v in d.get("latency",{}).items()})
print(d["extras"].get("warm_tick"), d["extras"].get("device_closed_loop"))
c=json.load(open("$OUT/bench_conv.json"))
print("CONV", c["metric"], c["value"], c["ms_per_step"], c["steps"], c["warmup"], c["roofline"]["frac"], c.get("error"), list(c.get("rti_pass",{}).keys()))
PY
python -m pytest tests/test_acado_compat.py tests/test_bench_json.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
