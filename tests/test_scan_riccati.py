"""csrc/nmpc_scan.h -- the backward Riccati recursion as an associative scan (what the (16, 2) / (16, 4) / (32, 1) mappings of the stage-block
kernel run instead of 20 .. 52 sequential stage steps): the element construction and the combine rule, compiled for the CPU, against the
sequential float64 recursion; on the GPU the scanned sweep against the sequential one (ALORE_NMPC_SCAN=0) on every mapping that has it.
Reference semantics: acado_solver.c:327-363 (the condensed QP whose stage-wise form both sweeps solve)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alore_legged_manipulator_amd", "csrc")


def test_scan_elements_and_combine_rule_on_the_cpu(tmp_path):
    exe = str(tmp_path / "scan_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", CSRC, os.path.join(ROOT, "tests", "harness", "scan_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    worst = float(r.stdout.strip().split()[-1])
    assert worst < 5e-5, r.stdout   # measured 6.9e-6


@pytest.mark.gpu
@pytest.mark.parametrize("N,lanes,B", [(20, 0x110, 300), (20, 0x120, 300), (50, 0x110, 200), (32, 0x110, 64), (32, 0x120, 64), (64, 0x110, 64), (20, 0x110, 4096)])
def test_scanned_backward_sweep_agrees_with_the_sequential_one(N, lanes, B, tmp_path):
    """Two ticks (cold: prediction + sweeps; warm: the dual names the working set) and five iterations inside one launch with the scan on
    and off (child processes: the switch is read once): same status everywhere; x and u within 2e-5 on the Monte-Carlo batch of the
    bench and within 5e-4 on the stress distribution (cond(H) ~ 2e3: two float32 solves of the same QP differ by up to 2e-4 there,
    measured; the hard cap of the parity rule -- each sweep is pinned against the oracle AND float64 by
    tests/test_gpu_parity.py::test_stage_block_kernel_on_the_stress_distribution on these mappings).  N = L * S: node N has no slot of
    its own and the terminal element goes behind the last lane's block."""
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch, make_wide_batch
B, N = {B}, {N}
out = {{}}
for dist, batch in (("bench", make_batch(B, N, seed=5, fast_tail=0.3)), ("stress", make_wide_batch(B, N, 77))):
    eng = BatchedNmpc(B, N, lanes_per_problem={lanes})
    eng.load(batch); eng.rti(1); a = eng.fetch(); eng.rti(1); b = eng.fetch()
    eng2 = BatchedNmpc(B, N, lanes_per_problem={lanes})
    eng2.load(batch); eng2.rti(5); c = eng2.fetch()
    assert eng.launch_info()["lanes_per_problem"] == {lanes}
    for tag, o in (("cold", a), ("warm", b), ("five", c)):
        for k in ("x", "u", "status", "n_iter"):
            out[dist + "_" + tag + "_" + k] = o[k]
np.savez(sys.argv[1], **out)
"""
    res = {}
    for tag, env in (("scan", {}), ("seq", {"ALORE_NMPC_SCAN": "0"})):
        path = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(path)
    for dist, tol in (("bench", 2e-5), ("stress", 5e-4)):
        for t in ("cold", "warm", "five"):
            assert np.array_equal(res["scan"][f"{dist}_{t}_status"], res["seq"][f"{dist}_{t}_status"]), (dist, t)
            ok = res["scan"][f"{dist}_{t}_status"] == 0
            assert ok.mean() > 0.9
            for k in ("x", "u"):
                a, b = res["scan"][f"{dist}_{t}_{k}"][ok].reshape(ok.sum(), -1), res["seq"][f"{dist}_{t}_{k}"][ok].reshape(ok.sum(), -1)
                err = np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))
                assert err < tol, (dist, t, k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("N,lanes", [(17, 0x110), (21, 0x110), (22, 0x110), (30, 0x110), (31, 0x120), (23, 0x120), (33, 0x110), (47, 0x110), (48, 0x110),
                                     (49, 0x110), (63, 0x110), (13, 0x120), (16, 0x110)])
def test_scanned_sweep_at_every_position_of_the_terminal_node(N, lanes):
    """Horizons that put node N at every kind of slot -- the first slot of a lane (N a multiple of S), the middle of a block, the last lane
    of the group, one short of L * S -- on the mappings whose backward sweep is a scan: a cold and a warm tick against the oracle
    (x, u <= 1e-4 relative: BASELINE.json; objective 1e-4)."""
    sys.path.insert(0, ROOT)
    from alore_legged_manipulator_amd.nmpc import BatchedNmpc
    from alore_legged_manipulator_amd.scenarios import make_batch, problem
    from oracle.drivers import Oracle
    B = 21
    batch = make_batch(B, N, seed=77 + N, fast_tail=0.4)
    eng = BatchedNmpc(B, N, lanes_per_problem=lanes)
    eng.load(batch)
    eng.rti(1)
    a = eng.fetch()
    assert eng.launch_info()["lanes_per_problem"] == lanes
    eng.rti(1)
    b = eng.fetch()
    orc = Oracle(N)
    for p in range(0, B, 2):
        orc.reset(); orc.initialize_solver(); orc.load(problem(batch, p))
        for out in (a, b):
            orc.preparation_step()
            assert orc.feedback_step() == 0 and out["status"][p] == 0
            for k in ("x", "u"):
                ref = orc.v[k]
                err = float(np.max(np.abs(out[k][p].reshape(-1) - ref)) / max(1.0, float(np.max(np.abs(ref)))))
                assert err < 1e-4, (N, p, k, err)
            assert abs(out["obj"][p] - orc.get_objective()) <= 1e-4 * max(1.0, abs(orc.get_objective()))
