import sys, numpy as np, torch
sys.path.insert(0, ".")
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
B, N = 8, 20
b = make_batch(B, N, seed=3, fast_tail=1.0)
b["y"][2, 5, 0] = np.nan
b["x0"][4, 1] = np.inf
b["W"][6, 3, 3, 3] = np.nan
e = BatchedNmpc(B, N); e.load(b); e.rti(1); o = e.fetch()
print("status", o["status"], "n_iter", o["n_iter"])
print("finite rows:", [bool(np.isfinite(o["u"][i]).all()) for i in range(B)])
e.rti(3); o = e.fetch(); print("status after 3 more", o["status"])
