cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/g8_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g8_tests.log
tail -4 gpurun_out/g8_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/g8_bench_driver.json 2> gpurun_out/g8_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/g8_bench_driver.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"]*1e3, "frac", d["roofline"]["frac"], "steady", d["steady_state"]["ms_per_step"]*1e3, "in_order", d["in_order"]["ms_per_step"]*1e3)
print("warm_tick", d["extras"]["warm_tick"])
print("errors:", [k for k,v in d["extras"].items() if isinstance(v, dict) and "error" in v])
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"))
PY
