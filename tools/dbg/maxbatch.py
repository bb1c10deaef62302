"""One-off: the whole 8-GPU batch of BASELINE configs[3] (262144 problems) on ONE GPU, checked against the oracle
at both ends of the batch and through batch-size independence (problem b gives the same answer in any batch)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from alore_legged_manipulator_amd.nmpc import BatchedNmpc
from alore_legged_manipulator_amd.scenarios import make_batch, problem
from oracle.drivers import Oracle
N, B = 20, 262144
t = time.time(); hb = make_batch(B, N); print("make_batch", round(time.time() - t, 1), "s")
e = BatchedNmpc(B, N); e.load(hb); e.rti(1); torch.cuda.synchronize()
c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
e.load(hb); torch.cuda.synchronize(); c0.record(); e.rti(1); c1.record(); torch.cuda.synchronize()
ms = c0.elapsed_time(c1)
out = e.fetch()
print(f"B={B}: {ms*1e3:.0f} us  {B/ms/1e3:.3e} solves/s  hbm {4192*B/ms/1e6/8000*100:.1f}%  status!=0: {(out['status']!=0).sum()}")
orc = Oracle(N)
worst = 0.0
for b in list(range(8)) + list(range(B - 8, B)) + [B // 2, B // 3]:
    orc.reset(); orc.initialize_solver(); orc.load(problem(hb, b)); orc.preparation_step(); assert orc.feedback_step() == 0
    worst = max(worst, float(np.max(np.abs(out["u"][b].reshape(-1) - orc.v["u"])) / max(1.0, np.max(np.abs(orc.v["u"])))))
print("worst rel err of u vs oracle on 18 sampled problems:", worst)
small = BatchedNmpc(4096, N); sb = {k: v[B - 4096:] for k, v in hb.items()}; small.load(sb); small.rti(1); so = small.fetch()
print("tail 4096 problems identical to a 4096-batch run:", bool(np.array_equal(so["u"], out["u"][B - 4096:]) and np.array_equal(so["x"], out["x"][B - 4096:])))
