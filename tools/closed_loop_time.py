#!/usr/bin/env python3
"""Time per tick of the device-resident closed loop (bench.py's device_closed_loop leg alone): B robots, 200 ticks.

    python3 tools/closed_loop_time.py [robots]
Environment (experiments): ALORE_NMPC_CLOSED_LOOP_SERIAL=1 (library: the three kernels of a tick in a row), CL_LANES (lanes_per_problem of
the engine, e.g. 272 = the (16, 2) stage-block mapping), CL_PG (warm_start_steps), CL_SHARED=1 (W / bounds / od as one copy), CL_STREAM=1
(a torch stream of its own instead of the default stream)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from alore_legged_manipulator_amd.scenarios import make_batch
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.host import Polynome

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N = 20
    dev = torch.device("cuda:0")
    batch = make_batch(B, N)
    rng = np.random.default_rng(7)
    vw = rng.uniform([0.5, -1.0], [1.8, 1.0], (B, 2))
    T = np.array([1.0, 1.0, 1.0]); Tc = np.cumsum(T)
    msgs = [Polynome(np.stack([w * Tc[:-1], v * Tc[:-1]], 1), T, [0, 0, w, v, 0, 0], [w * Tc[-1], v * Tc[-1], w, v, 0, 0],
                     [0, 0, 0], [-0.3, 0.3, 0.1], 0.0) for v, w in vw]
    e7 = BatchedNmpc(B, N, device=0, diagnostics=False, lanes_per_problem=int(os.environ.get('CL_LANES', '0')), warm_start_steps=int(os.environ.get('CL_PG', '-1')))
    e7.load({k: batch[k] for k in ("W", "WN", "lbValues", "ubValues")})
    if os.environ.get("CL_SHARED"):   # one robot class: W / WN, the bounds and od are one copy (alore_nmpc_set_shared_members)
        e7.set_shared_members(W=True, bounds=True, od=True)
    e7.refs_init(max_pieces=4, max_checkpoints=40)
    e7.refs_set_polynomes(np.arange(B), msgs)
    e7.plant_init()
    side = torch.cuda.Stream(device=dev) if os.environ.get("CL_STREAM") else None
    if side is not None:
        torch.cuda.synchronize(dev)
        torch.cuda.set_stream(side)
    for rep in range(3):
        e7.plant_set_state(np.zeros((B, 3)), np.tile([0.1, -0.3, 0.3], (B, 1)))
        e7.closed_loop_reset()
        for t in range(20):
            e7.closed_loop_tick(0.01 * (t + 1))
        torch.cuda.synchronize(dev)
        nt = 200
        t_a = time.perf_counter()
        e7.closed_loop_run(0.01 * 21, 0.01, nt)
        torch.cuda.synchronize(dev)
        t_b = time.perf_counter()
        pose, _, goal = e7.plant_get_state()
        print(f"B={B} ticks={nt} us_per_tick={(t_b - t_a) / nt * 1e6:.2f} unsolved={int((e7.t['status'] != 0).sum().item())} "
              f"finite={bool(np.isfinite(pose).all())} pose_sum={float(np.abs(pose).sum()):.6f}", flush=True)

if __name__ == "__main__":
    main()
