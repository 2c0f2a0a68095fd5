"""Numpy simulation of the eigen-free stage (csrc/letkf_krylov.hip) with the kernel's recurrences: column-wise single-reduction CG
on a bench workload's point matrices, the Lanczos tridiagonal from its coefficients, g_T(T_m) e_1 by a Chebyshev expansion on
the tridiagonal -- iteration counts and errors against an eigen-decomposition.  Usage: tools/sim_cg_lanczos.py WORKLOAD SPREAD [iid|correlated]"""
import sys, math, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench_workload as bw
np.set_printoptions(linewidth=200)
name = sys.argv[1]; spread = float(sys.argv[2]); kind = sys.argv[3] if len(sys.argv)>3 else 'correlated'
dev = torch.device('cpu')
w = bw.build(name, dev, ensval_kind=kind)
k, nv, npts = w['k'], w['nv'], w['npts']
gv = bw.state_view(w, w['gues'])
gv[:, :k] -= gv[:, :k].mean(dim=1, keepdim=True)
if kind=='correlated': bw.correlate_ensval(w)
ens = w['ensval'][:, :k]
sd = float(ens.std()); err = w['cfg']['err']
ens = ens * (spread*err/sd)
rng = np.random.default_rng(0)
pts = rng.choice(npts, 5, replace=False)
c = k-1.0

def cheb_coef(f, lo, hi, N):
    j = np.arange(N); x = np.cos(np.pi*(j+0.5)/N); L = 0.5*(hi-lo)*x + 0.5*(hi+lo)
    fv = f(L); i = np.arange(N)[:,None]
    cc = np.cos(np.pi*i*(j[None,:]+0.5)/N)
    cf = (cc*fv[None,:]).sum(1)*2.0/N; cf[0]*=0.5
    return cf
def cheb_apply_tri(a, b, f, lo, hi, tol=1e-17):
    # y = f(T) e1 for symmetric tridiagonal T (diag a, offdiag b), via Chebyshev on [lo,hi]
    m = len(a); cond = hi/lo; sk=math.sqrt(cond); rate=(sk-1)/(sk+1)
    deg = int(math.ceil(math.log(tol)/math.log(rate)))+2
    cf = cheb_coef(f, lo, hi, deg+1)
    half=0.5*(hi-lo); mid=0.5*(hi+lo)
    def mv(v):
        o = a*v
        o[:-1] += b*v[1:]; o[1:] += b*v[:-1]
        return (o - mid*v)/half
    t0 = np.zeros(m); t0[0]=1.0
    y = cf[0]*t0
    t1 = mv(t0); y += cf[1]*t1
    for d in range(2, deg+1):
        t2 = 2*mv(t1) - t0; y += cf[d]*t2; t0, t1 = t1, t2
    return y, deg

def cg_block(M, Tm, fT, tol=1e-15, mmax=200, variant='cgcg'):
    # column-wise independent CG (Chronopoulos-Gear single-reduction variant), shared matvec
    n, nb = Tm.shape
    x = np.zeros_like(Tm); r = Tm.copy()
    p = np.zeros_like(Tm); q = np.zeros_like(Tm)
    rho_old = np.ones(nb); alpha_old = np.ones(nb)
    t2 = (Tm*Tm).sum(0)
    R = []; alphas=[]; betas=[]; rhos=[]
    active = np.ones(nb, bool)
    for j in range(mmax):
        wv = M@r
        rho = (r*r).sum(0); mu = (r*wv).sum(0)
        R.append(r.copy()); rhos.append(rho.copy())
        if j==0:
            beta = np.zeros(nb); alpha = rho/mu
        else:
            beta = rho/rho_old; alpha = rho/(mu - rho*beta/alpha_old)
        betas.append(beta.copy()); alphas.append(alpha.copy())
        p = r + beta*p; q = wv + beta*q
        x = x + alpha*p; r = r - alpha*q
        rho_old = rho; alpha_old = alpha
        if np.all(((r*r).sum(0)) <= tol*tol*t2): break
    m = len(R)
    al = np.array(alphas); be = np.array(betas); rh = np.array(rhos)
    # Lanczos tridiagonal per column
    XT = np.zeros_like(Tm); degs=[]
    for b in range(nb):
        a = 1.0/al[:,b]; a[1:] += be[1:,b]/al[:-1,b]
        off = np.sqrt(be[1:,b])/al[:-1,b]
        g = np.abs(a).copy(); g[:-1]+=np.abs(off); g[1:]+=np.abs(off)
        hi = g.max()*(1+1e-12); lo = c*(1-1e-9)
        y, dg = cheb_apply_tri(a, off, fT, lo, hi); degs.append(dg)
        nrm = np.sqrt(rh[:,b]); sgn = (-1.0)**np.arange(m)
        # v_j = sgn_j r_j/|r_j| ; x = |t| sum_j y_j v_j
        coef = y*sgn/nrm*np.sqrt(t2[b])
        XT[:,b] = sum(coef[j]*R[j][:,b] for j in range(m))
    return x, XT, m, max(degs)

sqk=math.sqrt(k-1.0); sqc=math.sqrt(c)
for p_ in pts:
    o0, o1 = int(w['obs_off'][p_]), int(w['obs_off'][p_+1])
    idx = w['obs_idx'][o0:o1].long()
    wgt = (1.0/w['rdiag'][o0:o1]).numpy()
    Y = ens[idx].numpy(); Z = Y*np.sqrt(wgt)[:,None]
    n = Z.shape[0]
    X = gv[:, :k, p_].numpy().T  # k x nv
    dual = n < k
    if dual:
        M = Z@Z.T + c*np.eye(n); Tm = np.concatenate([np.sqrt(wgt)[:,None]*rng.standard_normal((n,2))*3, Z@X],1)
        fT = lambda L: -sqk/(sqc*np.sqrt(L)*(sqc+np.sqrt(L)))
    else:
        M = Z.T@Z + c*np.eye(k); Tm = np.concatenate([Z.T@(np.sqrt(wgt)[:,None]*rng.standard_normal((n,2))*3), X],1)
        fT = lambda L: sqk/np.sqrt(L)
    ev, U = np.linalg.eigh(M)
    xe = U@((U.T@Tm)/ev[:,None]); xte = U@((U.T@Tm)*fT(ev)[:,None])
    x, XT, m, dg = cg_block(M, Tm, fT)
    e1 = np.abs(x-xe).max(0)/np.abs(xe).max(0); e2 = np.abs(XT-xte).max(0)/np.abs(xte).max(0)
    sk=math.sqrt(ev[-1]/c); chdeg = math.log(1e-16)/math.log((sk-1)/(sk+1))
    print(f'n={n} cond={ev[-1]/c:7.1f} cheb_deg_tight={chdeg:5.0f} cg_iters={m} tri_deg={dg} err_inv={e1.max():.1e} err_T={e2.max():.1e}')
