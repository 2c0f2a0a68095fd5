#!/usr/bin/env python3
"""numpy simulation of the staged path's eigen stage (odd-even one-sided Jacobi, rotate-and-swap, the stop rules of
csrc/letkf_jacobi_dev.h) on observation-space matrices M = Z Z^T + c I of the C3 / C5 shape: does a preconditioned start
need fewer sweeps?  Compared: the columns of M (what the kernel does), of its Cholesky factor L (Veselic-Hari / Drmac),
of L^T, of L after sorting the diagonal, smaller shifts c, and the one-sided iteration on Z^T itself.
Result: no -- c = (k-1)/rho keeps cond(M) at ~1.6, the spectrum is one cluster, and every variant needs
8.0-8.4 sweeps at order 200 / k = 320 (7.1-7.9 with the sorted factor, which would cost a pivoted Cholesky per point)."""
import numpy as np, scipy.linalg as sl
rng=np.random.default_rng(1)
def jacobi_sweeps(G, maxsw=40):
    G=G.copy(); m=G.shape[1]
    if m%2: G=np.hstack([G,np.zeros((G.shape[0],1))]); m+=1
    nsteps=0; clean=0  # count of consecutive steps with no notconv; stop when a full cycle (m steps) is clean
    clean2=0
    while nsteps < maxsw*m:
        for parity in (0,1):
            i=np.arange(parity,m-1,2); j=i+1
            A=G[:,i]; B=G[:,j]
            a=(A*A).sum(0); b=(B*B).sum(0); g=(A*B).sum(0)
            ok=(a>0)&(b>0)
            g2=g*g; ab=a*b
            notconv=(ok&(g2>1e-20*ab)).any()
            rot=ok&(g2>1e-30*ab)
            d=b-a; hh=np.sqrt(d*d+4*g2)
            with np.errstate(all='ignore'):
                t=np.where(rot, 2*g*np.copysign(1.0,d)/(np.abs(d)+hh),0.0)
            notconv2=(ok&((g2>1e-16*ab)|(t*t>1e-12))).any()
            c=1/np.sqrt(1+t*t); s=t*c
            # rotate and swap:  new lower = the larger?  (just rotate then swap positions)
            An=c*A - s*B; Bn=s*A + c*B
            G[:,i]=Bn; G[:,j]=An
            nsteps+=1
            clean = 0 if notconv else clean+1
            clean2 = 0 if notconv2 else clean2+1
            if clean>=m or clean2>=m: return nsteps/m, G
    return nsteps/m, G
def make(n,k,err=3.0,rho=1.0):
    Y=rng.standard_normal((n,k))*2.0; Y-=Y.mean(1,keepdims=True)
    rloc=np.exp(-0.5*rng.uniform(0,1,n)*12)  # nd2 up to ~ (2*sqrt(10/3))^2=13.3
    w=rloc/(err*err)
    Z=np.sqrt(w)[:,None]*Y
    c=(k-1)/rho
    return Z@Z.T+c*np.eye(n)
for (n,k) in [(200,320),(200,1000),(110,320)]:
    res={}
    for trial in range(3):
        M=make(n,k)
        s0,G=jacobi_sweeps(M)
        L=np.linalg.cholesky(M)
        s1,G1=jacobi_sweeps(L)
        s2,G2=jacobi_sweeps(L.T.copy())
        # pivoted
        Lp,piv,_=sl.lapack.dpstrf(M,lower=1)[:3] if False else (None,None,None)
        dperm=np.argsort(-np.diag(M)); Mp=M[np.ix_(dperm,dperm)]; Lp=np.linalg.cholesky(Mp)
        s3,G3=jacobi_sweeps(Lp)
        ev=np.sort(np.linalg.eigvalsh(M)); e1=np.sort((G1*G1).sum(0))[-n:]
        print(n,k,"M:",round(s0,2),"L:",round(s1,2),"LT:",round(s2,2),"Lsorted:",round(s3,2),"eig err",np.abs(e1-ev).max()/ev.max(), "cond",ev[-1]/ev[0])
print("shift experiment")
def makeZ(n,k,err=3.0):
    Y=rng.standard_normal((n,k))*2.0; Y-=Y.mean(1,keepdims=True)
    rloc=np.exp(-0.5*rng.uniform(0,1,n)*12)
    return np.sqrt(rloc/(err*err))[:,None]*Y
for (n,k) in [(200,320),(200,1000)]:
    Z=makeZ(n,k); S=Z@Z.T; c=k-1.0
    ev=np.linalg.eigvalsh(S); print("S eig range",ev[0],ev[-1],"c",c)
    for s in [c, 0.3*c, 0.1*c, 0.03*c, 0.01*c, 0.0]:
        sw,G=jacobi_sweeps(S+s*np.eye(n))
        lam=np.sqrt((G*G).sum(0))[:n]
        print(n,k,"shift",round(s,2),"sweeps",round(sw,2),"eig err",np.abs(np.sort(lam)-np.sort(ev+s)).max()/c)
    sw,G=jacobi_sweeps(Z.T.copy()); print("one-sided on Z^T: sweeps",round(sw,2))
