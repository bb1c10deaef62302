import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle.ltv_mpc_oracle import *
FREE, BLO, BHI, RLO, RHI = 0,1,2,3,4

def ws_riccati(xbar, xref, dref, p, max_sweeps=64, verbose=False, single_after=24):
    d, K = p.delay_num, p.T - p.delay_num
    Qp2 = 2*np.array([p.Q[0], p.Q[1], p.Q[3]])
    A=[];B=[];C=[]
    for j in range(K):
        a,b,c = linear_model(xbar[d+j], p); A.append(a);B.append(b);C.append(c)
    pos0 = xbar[d][:3].copy()
    umax = np.array([p.max_speed, p.max_omega]); rmax = np.array([p.max_cv, p.max_comega])
    Rd2 = 2*np.array(p.Rd); Ruu0 = 2*np.array([p.R[0]+p.Q[2], p.R[1]])
    st = np.zeros((K,2), int)
    tol=1e-9
    for sweep in range(max_sweeps):
        # effective boxes: a RATE-linked stage hands its box on to its predecessor, shifted by the rate limit
        hi_eff=np.tile(umax,(K,1)); lo_eff=np.tile(-umax,(K,1))
        for j in range(K-1,0,-1):
            for c in range(2):
                if st[j,c]==RHI: hi_eff[j-1,c]=min(umax[c], hi_eff[j,c]-rmax[c])
                if st[j,c]==RLO: lo_eff[j-1,c]=max(-umax[c], lo_eff[j,c]+rmax[c])
        # backward
        P = np.zeros((5,5)); pv=np.zeros(5)
        rec=[None]*K
        for j in range(K-1,-1,-1):
            # quadratic form in w = (xi(5), u0, u1): next state xi' = Fw w + c
            Fw = np.zeros((5,7)); Fw[:3,:3]=A[j]; Fw[:3,5:7]=B[j]; Fw[3,5]=1; Fw[4,6]=1
            c5=np.zeros(5); c5[:3]=C[j]
            Pt = P.copy(); Pt[:3,:3]+=np.diag(Qp2)
            pt = pv.copy(); pt[:3] -= Qp2*xref[:,d+j]
            H = Fw.T@Pt@Fw; h = Fw.T@(Pt@c5+pt)
            H[5,5]+=Ruu0[0]; H[6,6]+=Ruu0[1]; h[5]+= -2*p.Q[2]*dref[0,d+j]
            if j>=1:
                for c in range(2):
                    # Rd (u_c - prev_c)^2 -> Hessian 2Rd on (u_c - prev_c)
                    iu, ip = 5+c, 3+c
                    H[iu,iu]+=Rd2[c]; H[ip,ip]+=Rd2[c]; H[iu,ip]-=Rd2[c]; H[ip,iu]-=Rd2[c]
            # eliminate u1 (index 6) then u0 (index 5)
            info=[]
            idx=list(range(7))
            for k,c in ((6,1),(5,0)):
                pos = idx.index(k)
                rest=[i for i in range(len(idx)) if i!=pos]
                grad_row = (H[pos,:].copy(), h[pos], list(idx))   # gradient wrt u_c: row . w + h
                s=st[j,c]
                a=np.zeros(len(rest)); f=0.0
                if s==FREE:
                    a = -H[pos,rest]/H[pos,pos]; f = -h[pos]/H[pos,pos]
                elif s in (BLO,BHI):
                    f = lo_eff[j,c] if s==BLO else hi_eff[j,c]
                else:
                    a[ [idx[i] for i in rest].index(3+c) ] = 1.0; f = -rmax[c] if s==RLO else rmax[c]
                Hrr=H[np.ix_(rest,rest)]; Hkr=H[pos,rest]; Hkk=H[pos,pos]
                Hn = Hrr + np.outer(a,Hkr)+np.outer(Hkr,a)+Hkk*np.outer(a,a)
                hn = h[rest] + a*h[pos] + (Hkr + a*Hkk)*f
                info.append((c, a, f, [idx[i] for i in rest], grad_row))
                H,h = Hn,hn; idx=[idx[i] for i in rest]
            P,pv = H,h
            rec[j]=info
        # forward
        xi=np.zeros(5); xi[:3]=pos0
        u=np.zeros((K,2)); newst=st.copy(); nchg=0; chain_mu=np.zeros(2); chain_dir=np.zeros(2,int); cands=[]
        for j in range(K):
            info=rec[j]
            # u0 first (second eliminated), depends on xi only
            c0,a0,f0,vars0,g0 = info[1]
            w = {i:xi[i] for i in range(5)}
            u0 = sum(a0[t]*w[vars0[t]] for t in range(len(vars0))) + f0
            w[5]=u0
            c1,a1,f1,vars1,g1 = info[0]
            u1 = sum(a1[t]*w[vars1[t]] for t in range(len(vars1))) + f1
            w[6]=u1
            uu=[u0,u1]
            for (c,a,f,vars_,g) in info:
                row,hh,ids = g
                grad = sum(row[t]*w[ids[t]] for t in range(len(ids))) + hh
                s=st[j,c]; val=uu[c]; prev=xi[3+c]
                ns=s; sev=0.0
                if s==FREE:
                    chain_mu[c]=0.0; chain_dir[c]=0
                    vb = max(lo_eff[j,c]-val, val-hi_eff[j,c]); vr = max(-rmax[c]-(val-prev), (val-prev)-rmax[c]) if j>=1 else -1
                    if vb>tol and vb>=vr: ns = BLO if val<lo_eff[j,c] else BHI; sev=vb
                    elif vr>tol: ns = RLO if (val-prev)<0 else RHI; sev=vr
                elif s in (BLO,BHI):
                    lower = s==BLO
                    tightened = (lo_eff[j,c] > -umax[c]+1e-12) if lower else (hi_eff[j,c] < umax[c]-1e-12)
                    chain_mu[c] = grad if tightened else 0.0   # = -(+-mu_b): the multiplier of the box at the end of the chain
                    chain_dir[c] = (-1 if lower else 1) if tightened else 0
                    if (lower and grad < -tol) or ((not lower) and grad > tol): ns=FREE; sev=abs(grad)
                    elif j>=1 and abs(val-prev)>rmax[c]+tol: ns = RLO if val-prev<0 else RHI; sev=abs(val-prev)-rmax[c]
                else:
                    lower = s==RLO
                    if chain_dir[c] != (-1 if lower else 1): chain_mu[c]=0.0; chain_dir[c]=0
                    g_eff = grad - chain_mu[c]
                    if (lower and g_eff < -tol) or ((not lower) and g_eff > tol):
                        # inside a chain that hangs from a box further down: this stage becomes the anchor of what remains
                        ns = FREE if chain_mu[c]==0.0 else (BLO if lower else BHI); sev=abs(g_eff)
                        chain_mu[c] = 0.0; chain_dir[c]=0
                if ns!=s: nchg+=1; cands.append((sev,j,c,ns))
                newst[j,c]=ns
            u[j]=uu
            nxt=np.zeros(5); nxt[:3]=A[j]@xi[:3]+B[j]@u[j]+C[j]; nxt[3:]=u[j]; xi=nxt
        if verbose: print(" sweep",sweep,"changes",nchg)
        if sweep>=single_after and cands:
            newst=st.copy(); _,j,c,ns=max(cands); newst[j,c]=ns
        st=newst
        if nchg==0: return u, sweep+1, True
    return u, max_sweeps, False

