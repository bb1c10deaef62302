#!/usr/bin/env python3
"""Two-phase grids on WARM ticks: the second and third real-time iteration of the same batches (the dual of the last tick names the working set,
as in a fleet that tracks its trajectories) -- one-pass grid against two-phase grid, time of the grid by the library's events and bits.
usage: two_phase_warm.py [slots]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alore_legged_manipulator_amd.nmpc import BatchedNmpc  # noqa: E402
from alore_legged_manipulator_amd.scenarios import make_batch  # noqa: E402

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N = 4096, 20
batch = make_batch(B, N)
res = {}
for mode in (0, 1):
    eng = BatchedNmpc(B, N, slots=slots)
    eng.set_two_phase(mode)
    times = {1: [], 2: [], 3: []}
    for rep in range(4):
        eng.load(batch, slot=None)
        torch.cuda.synchronize()
        eng.set_timing(True)
        for tick in (1, 2, 3):
            eng.rti_range(0, slots)
            torch.cuda.synchronize()
            times[tick].append(float(eng.launch_info()["last_kernel_ms"]) * 1e3)
            if rep == 0:
                res[(mode, tick)] = {k: eng.ts[k].clone() for k in ("x", "u", "dual", "status", "kkt", "obj")}
                if mode == 1:
                    res[("share", tick)] = eng.two_phase_info()["tail_share"]
        eng.set_timing(False)
    for tick in (1, 2, 3):
        print(f"{'two phases' if mode else 'one pass  '} tick {tick}: grid {min(times[tick]):.1f} us ({min(times[tick]) / slots:.2f} us per batch)"
              + (f", share of the problems queued {res[('share', tick)]:.4f}" if mode else ""))
for tick in (1, 2, 3):
    same = all(bool(torch.equal(res[(0, tick)][k], res[(1, tick)][k])) for k in res[(0, tick)])
    print(f"tick {tick}: x, u, dual, status, kkt, obj bit-equal: {same}")
    if not same:
        for k in ("x", "u", "dual", "status", "kkt", "obj"):
            a, b = res[(0, tick)][k].double(), res[(1, tick)][k].double()
            d = (a - b).abs().reshape(a.shape[0], a.shape[1], -1).amax(dim=2)
            if k == "u":
                print("      per batch, problems whose u differs:", [int(v) for v in (d > 0).sum(dim=1)])
                n1 = res[(0, tick)]["status"]
            print(f"      {k:6s}: problems that differ {int((d > 0).sum())} of {d.numel()}, max abs difference {float(d.max()):.3e}")
