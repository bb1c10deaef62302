set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_ltv_mpc.py tests/test_edge_cases_gpu.py -m gpu -q > gpurun_out/g1_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g1_tests.log
tail -5 gpurun_out/g1_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/g1_bench_driver.json 2> gpurun_out/g1_bench_driver.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-graph > gpurun_out/g1_bench_driver_nograph.json 2>/dev/null
timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/g1_bench_default.json 2> gpurun_out/g1_bench_default.err
timeout 600 python bench.py --many-mode streams --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/g1_bench_streams.json 2>/dev/null
python - <<'PY'
import json
for f in ("g1_bench_driver","g1_bench_driver_nograph","g1_bench_default","g1_bench_streams"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
        print(f, "ms_per_step", d["ms_per_step"]*1e3, "kernel", d["roofline"]["kernel_ms_avg"]*1e3, "frac", d["roofline"]["frac"], "steady", d.get("steady_state",{}).get("ms_per_step"), "in_order", d["in_order"]["ms_per_step"], d.get("parity_spot_check",{}).get("worst_rel"))
    except Exception as e:
        print(f, "ERR", e)
PY
