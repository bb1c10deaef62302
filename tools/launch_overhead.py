"""Where the time of the driver's 20-step region goes outside the grid: the same rti_range(warmup, K) call replayed from a
one-node hipGraph and launched eagerly, host clock (sync -> call -> sync) and HIP events around the call, 40 repetitions each.

    python tools/launch_overhead.py [K]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alore_legged_manipulator_amd.nmpc import BatchedNmpc  # noqa: E402
from alore_legged_manipulator_amd.scenarios import make_batch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N, W = 4096, 20, 5
dev = torch.device("cuda", 0)
batch = make_batch(B, N)
eng = BatchedNmpc(B, N, device=0, slots=K + W)
eng.set_launch_overlap(16)
eng.load(batch, slot=None)
keep = {k: eng.ts[k].clone() for k in ("x", "u", "dual")}
eng.rti_range(0, W)
torch.cuda.synchronize()

side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        eng.rti_range(W, K)
torch.cuda.current_stream(dev).wait_stream(side)


def run(fn, reps=40):
    host, evs = [], []
    for _ in range(reps):
        for k in keep:
            eng.ts[k].copy_(keep[k])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        t_call = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        host.append(((t1 - t0) * 1e6, (t_call - t0) * 1e6))
        evs.append(e0.elapsed_time(e1) * 1e3)
    h = np.array(host)
    return np.median(h[:, 0]), np.min(h[:, 0]), np.median(h[:, 1]), np.median(evs), np.min(evs)


for name, fn in (("graph replay", g.replay), ("eager rti_range", lambda: eng.rti_range(W, K)), ("graph replay", g.replay),
                 ("eager rti_range", lambda: eng.rti_range(W, K))):
    med, mn, call, ev, evmin = run(fn)
    print(f"{name:16s} K={K}: host sync->sync median {med:7.1f} us (min {mn:7.1f}; the call returns after {call:6.1f});  events median {ev:7.1f} us (min {evmin:7.1f})"
          f"  -> {med / K:.2f} us per step by the host clock, {ev / K:.2f} by events")
print(eng.launch_info())
