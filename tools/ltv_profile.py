#!/usr/bin/env python3
"""LTV-MPC, B robots x 5 relinearisations, cold then warm: for rocprofv3 kernel-trace / PMC passes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alore_legged_manipulator_amd.ltv_mpc import BatchedLtvMpc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(11)
lt = BatchedLtvMpc(B)
T = lt.cfg.predict_steps
vv, ww = rng.uniform(0.5, 2.5, B), rng.uniform(-1.5, 1.5, B)
ts = (np.arange(T) + 1) * lt.cfg.dt
xr = np.stack([vv[:, None] / ww[:, None] * np.sin(ww[:, None] * ts), vv[:, None] / ww[:, None] * (1 - np.cos(ww[:, None] * ts)), ww[:, None] * ts], 2)
dr = np.stack([np.repeat(vv[:, None], T, 1), np.repeat(ww[:, None], T, 1)], 2)
st0 = rng.uniform(-0.1, 0.1, (B, 3))
lt.set_refs(xr, dr)
t0 = time.perf_counter(); g0 = lt.get_cmd(st0, n_relin=5, reset=True); t1 = time.perf_counter()
for i in range(5):
    g1 = lt.get_cmd(st0, n_relin=5)
t2 = time.perf_counter()
print(f"B={B}: cold {1e3 * (t1 - t0):.2f} ms, warm {1e3 * (t2 - t1) / 5:.2f} ms per tick (5 relinearisations), unsettled {int((g1['status'] != 0).sum())}")
# tick entry point (what a controller calls): states up, one launch, 16 B per robot down
t3 = time.perf_counter()
for i in range(20):
    lt.tick(st0, n_relin=5)
t4 = time.perf_counter()
print(f"B={B}: alore_ltv_tick warm {1e3 * (t4 - t3) / 20:.3f} ms; sweeps cold mean {g0['sweeps'].mean():.2f} max {g0['sweeps'].max()}, warm mean {g1['sweeps'].mean():.2f}")
