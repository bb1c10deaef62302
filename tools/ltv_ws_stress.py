import numpy as np, sys, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/tmp/proto'); sys.path.insert(0,'/root/repo/tests')
from oracle.ltv_mpc_oracle import *
mod = importlib.import_module(sys.argv[1]); ws_riccati = mod.ws_riccati
from test_ltv_mpc import random_case
bad=0; worst=0; sws=[]
for delay in (0,1,3):
    p = LtvParams(delay_num=delay); rng = np.random.default_rng(20+delay)
    for i in range(40):
        now,out,buff,xref,dref = random_case(rng,p)
        xbar=predict_motion(now,out,p)
        u,sw,ok = ws_riccati(xbar,xref,dref,p); sws.append(sw)
        if i%4==0 or not ok:
            ref, z, info, xb = solve_mpcv(now,out,buff,xref,dref,p)
            e=np.max(np.abs(u.T-ref[:,p.delay_num:])); worst=max(worst,e)
            if not ok or e>1e-6: bad+=1; print("moderate delay",delay,"case",i,ok,sw,e)
print("moderate: bad",bad,"worst",worst,"sweeps mean",np.mean(sws),"max",max(sws))
rng = np.random.default_rng(999); bad=0; worst=0; sws=[]
for i in range(400):
    delay = int(rng.integers(0,4)); p = LtvParams(delay_num=delay)
    v, w = rng.uniform(-0.5, 3.5), rng.uniform(-3.5, 3.5)
    ts=(np.arange(p.T)+1)*p.dt
    if abs(w)<1e-3: w=0.1
    xref=np.stack([v/w*np.sin(w*ts), v/w*(1-np.cos(w*ts)), w*ts]); dref=np.stack([np.full(p.T,v),np.full(p.T,w)])
    now=[rng.uniform(-1,1),rng.uniform(-1,1),rng.uniform(-1.5,1.5),0.0]
    out=np.zeros((2,p.T)); out[0]=rng.uniform(-1,3.5)*rng.uniform(0,1,p.T); out[1]=rng.uniform(-3,3)
    buff=[np.array([rng.uniform(0,3),rng.uniform(-2,2)]) for _ in range(delay)]
    for k in range(delay): out[:,k]=buff[k]
    xbar = predict_motion(now,out,p)
    u,sw,ok = ws_riccati(xbar,xref,dref,p); sws.append(sw)
    if not ok: bad+=1; print("wide case",i,"delay",delay,"NOT settled"); continue
    if i%8==0 or i in (49,64,130,360):
        ref, z, info, xb = solve_mpcv(now,out,buff,xref,dref,p)
        e=np.max(np.abs(u.T-ref[:,p.delay_num:])); worst=max(worst,e)
        if e>1e-6: bad+=1; print("wide case",i,"err",e)
print("wide: bad",bad,"worst",worst,"sweeps mean",np.mean(sws),"max",max(sws), "p99", np.percentile(sws,99))
