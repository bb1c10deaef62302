import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from alore_legged_manipulator_amd.whole_body import BatchedWholeBody, model_info
from wb_cases import make_problems_fast, weights
B,N=4096,20
eng=BatchedWholeBody(B,N,0.01); eff=model_info()["effort"]
x0,xref,uref,xi,ui=make_problems_fast(B,N,seed=3)
eng.set_weights(*weights()); eng.set_problem(x0,xref,uref); eng.set_iterate(xi,ui)
for it in range(4):
    eng.rti(1); l,r=eng.last_times(); x,u=eng.get_iterate()
    sat=np.mean(np.any(np.abs(u[:,:,:18])>=eff-1e-9,axis=2))
    print(f"iteration {it}: linearise {l:.2f} ms riccati {r:.2f} ms, stages with a saturated torque {sat:.3f}")
