"""Diagnostic: distribution of working-set iterations over the bench batch vs. prediction steps."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
B, N = 4096, 20
batch = make_batch(B, N)
for pg in (0, 4, 6, 8, 10, 12, 16, 24):
    s = BatchedNmpc(B, N, warm_start_steps=pg)
    s.load(batch)
    s.rti(1)
    torch.cuda.synchronize()
    it = s.t["n_iter"].cpu().numpy()
    st = s.t["status"].cpu().numpy()
    h = np.bincount(it, minlength=8)[:8]
    waves = it.reshape(-1, 2).max(1)
    print(f"pg={pg:2d} n_iter hist {h.tolist()} mean {it.mean():.4f} bad {int((st != 0).sum())} waves(L=32) with >=2: {(waves >= 2).sum()} >=3: {(waves >= 3).sum()}")
    s.close()
