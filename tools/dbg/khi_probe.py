"""Diagnostic: for the bench problems that need a second working-set sweep, which stages change?"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
B, N = 4096, 20
batch = make_batch(B, N)
def run(max_it):
    s = BatchedNmpc(B, N, max_as_iter=max_it)
    s.load(batch); s.rti(1); torch.cuda.synchronize()
    o = {k: s.t[k].cpu().numpy().copy() for k in ("dual", "u", "n_iter", "status")}
    s.close(); return o
full = run(64); one = run(1)
idx = np.nonzero(full["n_iter"] >= 2)[0]
print("problems with >=2:", len(idx))
for b in idx:
    a_full = np.sign(full["dual"][b]).astype(int)          # [N,2] active set signs
    a_one = np.sign(one["dual"][b]).astype(int)
    ch = np.nonzero((a_full != a_one).any(1))[0]
    print(b, "n_iter", full["n_iter"][b], "changed stages", ch.tolist(), "active stages(final)", np.nonzero(a_full.any(1))[0].tolist())
