"""One batch at a time (B = 4096, (16, 2) mapping) and one problem (B = 1) against the number of working-set prediction steps:
microseconds per launch in a back-to-back eager loop (HIP events) and p50 of one synchronous launch (host clock)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alore_legged_manipulator_amd.nmpc import BatchedNmpc  # noqa: E402
from alore_legged_manipulator_amd.scenarios import make_batch  # noqa: E402

N = 20
for B in (4096, 1):
    batch = make_batch(B, N)
    for pg in (-1, 3, 4, 5, 6, 7, 8, 10):
        eng = BatchedNmpc(B, N, slots=8, warm_start_steps=pg)
        eng.load(batch, slot=None)
        keep = {k: eng.ts[k].clone() for k in ("x", "u", "dual")}
        for s in range(8):
            eng.rti(1, slot=s)
        torch.cuda.synchronize()
        for k in keep:
            eng.ts[k].copy_(keep[k])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(25):
            for s in range(8):
                eng.rti(1, slot=s)
        e1.record(); torch.cuda.synchronize()
        loop_us = e0.elapsed_time(e1) * 1e3 / 200
        nit = float(eng.ts["n_iter"].float().mean().item())
        ts = []
        for i in range(300):
            for k in keep:
                eng.ts[k][0].copy_(keep[k][0])
            torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.rti(1, slot=0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"B={B} prediction steps {pg:3d}: loop {loop_us:6.2f} us per launch (warm iterates after the first round), sync launch p50 {np.percentile(ts[50:], 50) * 1e6:6.2f} us, "
              f"lanes {eng.launch_info()['lanes_per_problem']:#x}")
        eng.close()
