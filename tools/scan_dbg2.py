import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch
B, N = int(sys.argv[1]), int(sys.argv[2])
lanes = int(sys.argv[3], 0)
batch = make_batch(B, N, seed=99, fast_tail=0.0)
def run(seq):
    eng = BatchedNmpc(B, N, lanes_per_problem=lanes)
    eng.load(batch)
    for k in seq: eng.rti(k)
    out = eng.fetch()
    return np.concatenate([out["x"].reshape(B,-1), out["u"].reshape(B,-1)], axis=1)
a = run([1, 1]); b = run([2])
d = np.abs(a - b).max(axis=1)
print("rti(1) x 2 vs rti(2): max diff", d.max(), "problems > 1e-4:", int((d > 1e-4).sum()), "scan", os.environ.get("ALORE_NMPC_SCAN", "1"), "stamps", os.environ.get("ALORE_NMPC_STAMPS", "0"))
c = run([1])
np.save("/tmp/one_%s_%s.npy" % (os.environ.get("ALORE_NMPC_SCAN", "1"), os.environ.get("ALORE_NMPC_STAMPS", "0")), c)
