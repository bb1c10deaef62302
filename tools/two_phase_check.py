#!/usr/bin/env python3
"""Two-phase grids of alore_nmpc_rti_many against the one-pass grid: same bits for x, u, dual, status, kkt, obj on the bench batch and on
the stress distribution, then the time of both (HIP events of the library around the grid).
usage: two_phase_check.py [slots] [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alore_legged_manipulator_amd.nmpc import BatchedNmpc  # noqa: E402
from alore_legged_manipulator_amd.scenarios import make_batch, make_wide_batch  # noqa: E402


def run(slots, B, N, batches, two_phase, reps=0):
    eng = BatchedNmpc(B, N, slots=slots)
    eng.set_two_phase(two_phase)
    for s in range(slots):
        eng.load(batches[s % len(batches)], slot=s)
    torch.cuda.synchronize()
    eng.rti_range(0, slots)
    torch.cuda.synchronize()
    out = {k: eng.ts[k].clone() for k in ("x", "u", "dual", "status", "n_iter", "kkt", "obj")}
    info = eng.two_phase_info()
    times = []
    if reps:
        eng.set_timing(True)
        for _ in range(reps):
            for s in range(slots):
                eng.load(batches[s % len(batches)], slot=s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.rti_range(0, slots)
            torch.cuda.synchronize()
            host = (time.perf_counter() - t0) * 1e6
            times.append((float(eng.launch_info()["last_kernel_ms"]) * 1e3, host))
        eng.set_timing(False)
        info = eng.two_phase_info()
    return out, info, times


def main():
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    N = 20
    for name, batches in (("bench batch", [make_batch(B, N)]),
                          ("mixed: bench / stress / fast tail", [make_batch(B, N, seed=100 + s, fast_tail=0.3) if s % 2 == 0 else make_wide_batch(B, N, 40 + s) for s in range(5)])):
        a, ia, ta = run(slots, B, N, batches, 0, reps=6)
        b, ib, tb = run(slots, B, N, batches, 1, reps=6)
        print(f"== {name}: {slots} slots of B = {B}")
        print("   one pass :", ia, "status != 0:", int((a["status"] != 0).sum()))
        print("   two phase:", ib, "status != 0:", int((b["status"] != 0).sum()))
        for k in ("x", "u", "dual", "status", "kkt", "obj"):
            same = torch.equal(a[k], b[k])
            extra = ""
            if not same:
                d = (a[k].double() - b[k].double()).abs()
                extra = f" max abs diff {float(d.max()):.3e}, differing elements {int((d > 0).sum())} of {d.numel()}"
            print(f"   {k:7s} bit-equal: {same}{extra}")
        ni_a, ni_b = a["n_iter"].float(), b["n_iter"].float()
        print(f"   n_iter mean one pass {float(ni_a.mean()):.3f}, two phase {float(ni_b.mean()):.3f}; two phase > one pass in {int((ni_b > ni_a).sum())} problems")
        print(f"   problems with more than one sweep (one pass): {float((ni_a > 1).float().mean()) * 100:.1f} %")
        fmt = lambda t: " ".join(f"{k:.1f}/{h:.1f}" for k, h in t)
        print(f"   grid us (events / host clock) one pass : {fmt(ta)}")
        print(f"   grid us (events / host clock) two phase: {fmt(tb)}")
        print(f"   per batch: one pass {min(k for k, _ in ta) / slots:.2f} us, two phase {min(k for k, _ in tb) / slots:.2f} us")


if __name__ == "__main__":
    main()
