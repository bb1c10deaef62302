import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
B, N = int(sys.argv[1]), int(sys.argv[2])
lanes = int(sys.argv[3], 0)
batch = make_batch(B, N, seed=99, fast_tail=0.0)
for K in (1, 2, 3, 15):
    eng = BatchedNmpc(B, N, lanes_per_problem=lanes)
    eng.load(batch); eng.rti(K); out = eng.fetch()
    np.save(f"/tmp/scan_dbg_{os.environ.get('ALORE_NMPC_SCAN','1')}_{K}.npy", np.concatenate([out["x"].reshape(B,-1), out["u"].reshape(B,-1)], axis=1))
    print(os.environ.get('ALORE_NMPC_SCAN','1'), K, eng.launch_info()["lanes_per_problem"], "n_iter mean", out["n_iter"].mean(), "status", (out["status"]!=0).sum())
