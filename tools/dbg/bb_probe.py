import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
B, N = 4096, 20
batch = make_batch(B, N)
s = BatchedNmpc(B, N); s.load(batch); s.rti(1); torch.cuda.synchronize()
it = s.t["n_iter"].cpu().numpy(); print(os.environ.get("ALORE_NMPC_PG_BB"), "hist", np.bincount(it, minlength=6)[:6].tolist(), "bad", int((s.t["status"]!=0).sum()))
