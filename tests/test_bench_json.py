"""bench.py's JSON line: no leg may report an error (a rebound variable silently turned three extras into
`{"error": ...}` and nobody looked at the final JSON).

CPU: the newest committed profiles/r*_bench.json must be clean and carry the contract's blocks.
GPU: a short bench.py run with every extra on must be clean."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def error_paths(obj, path=""):
    """every path in a nested JSON value at which a dict carries an `error` key"""
    found = []
    if isinstance(obj, dict):
        if "error" in obj:
            found.append(path or "/")
        for k, v in obj.items():
            found += error_paths(v, f"{path}/{k}")
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            found += error_paths(v, f"{path}/{i}")
    return found


def check_line(line: dict):
    assert error_paths(line) == []
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "config"):
        assert k in line, k
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert line["unsolved_problems"] == 0


def test_error_paths_finds_nested_errors():
    assert error_paths({"a": {"b": {"error": "x"}}, "c": [{"error": 1}]}) == ["/a/b", "/c/0"]
    assert error_paths({"a": 1, "b": {"c": [1, 2]}}) == []


def test_newest_committed_bench_json_is_clean():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_bench.json")))
    files = [f for f in files if os.path.basename(f) >= "r03"]   # rounds before the guard existed are history
    if not files:
        pytest.skip("no bench record of this round committed yet")
    line = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    check_line(line)
    if "extras" not in line or "cpu_baseline" not in line:
        # since round 4 the default-length line is recorded without the extras and / or the CPU baseline (`--no-extras`,
        # `--no-cpu-baseline`) and the full line under the driver's flags beside it: the full one carries the contract's blocks
        full = files[-1].replace("_bench.json", "_bench_driver_flags.json")
        assert os.path.exists(full), full
        line = json.loads(open(full).read().strip().splitlines()[-1])
        check_line(line)
    assert "extras" in line and "cpu_baseline" in line
    for leg in ("large_batch", "compact_io", "without_diagnostics", "reference_horizon_n50", "ltv_mpc",
                "backend_to_nmpc_pipeline", "whole_body_b2z1", "device_closed_loop", "controller_tick"):
        assert leg in line["extras"], leg


@pytest.mark.gpu
def test_short_bench_run_has_no_errors():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    check_line(line)
    assert line["steady_state"]["steps"] >= 200 and "extras" in line


def test_the_traffic_record_the_bench_line_quotes_is_valid_json():
    """bench.py reads profiles/hbm_traffic.json for roofline.traffic and swallows a parse error as 'no record': a
    hand-edited record with an unescaped quote silently turned the field into null (round 3)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = json.load(open(os.path.join(root, "profiles", "hbm_traffic.json")))
    for key, e in rec.items():
        assert e["bytes_per_launch"] >= e["algorithmic_bytes_per_launch"] > 0, key
        assert "kernel" in e and "source" in e, key
