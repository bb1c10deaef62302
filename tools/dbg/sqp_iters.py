"""Diagnostic: working-set iterations per SQP iteration of a cold-started converged solve (K = 15)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
B, N = 4096, 20
s = BatchedNmpc(B, N); s.load(make_batch(B, N))
for k in range(15):
    s.rti(1); torch.cuda.synchronize()
    it = s.t["n_iter"].cpu().numpy()
    print(f"sqp {k:2d}: n_iter mean {it.mean():.3f}  hist {np.bincount(it, minlength=6)[:6].tolist()}  waves with >=2: {(it.reshape(-1,2).max(1)>=2).sum()}  kkt max {float(s.t['kkt'].max()):.2e}")
