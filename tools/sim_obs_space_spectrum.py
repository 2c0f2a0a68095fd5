"""Spectrum of S = Z Z^T of sampled points of a bench workload at a given obs-space spread: dominant modes, the norm bound
against the true lambda_max, Chebyshev degrees with and without deflation.  Usage: tools/sim_obs_space_spectrum.py WORKLOAD SPREAD"""
import sys, math, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench_workload as bw
name = sys.argv[1]; spread = float(sys.argv[2])
dev = torch.device('cpu')
w = bw.build(name, dev, ensval_kind='correlated')
k, nv, npts = w['k'], w['nv'], w['npts']
gv = bw.state_view(w, w['gues'])
mean = gv[:, :k].mean(dim=1, keepdim=True)
gv[:, :k] -= mean
bw.correlate_ensval(w)
ens = w['ensval'][:, :k]
sd = float(ens.std())
err = w['cfg']['err']
print('sigma_o/err as built', sd/err)
ens = ens * (spread*err/sd)
rng = np.random.default_rng(0)
pts = rng.choice(npts, 40, replace=False)
c = k-1.0
res=[]
for p in pts:
    o0, o1 = int(w['obs_off'][p]), int(w['obs_off'][p+1])
    idx = w['obs_idx'][o0:o1].long()
    wgt = (1.0/w['rdiag'][o0:o1]).numpy()
    Y = ens[idx].numpy()
    Z = Y*np.sqrt(wgt)[:,None]
    n = Z.shape[0]
    S = Z@Z.T if n<k else Z.T@Z
    ev = np.linalg.eigvalsh(S)[::-1]
    fro = np.sqrt((S*S).sum()); inf = np.abs(S).sum(1).max()
    res.append((n, ev, min(fro,inf)))
for n, ev, b in res[:6]:
    print('n',n,'lmax',ev[0],'bound',b,'cond true',1+ev[0]/c,'cond bound',1+b/c)
    print('  top10/c', np.round(ev[:10]/c,2), ' ev[20]/c', round(ev[20]/c,2), 'ev[50]/c', round(ev[min(50,len(ev)-1)]/c,2))
ct = np.array([1+ev[0]/c for n,ev,b in res]); cb = np.array([1+b/c for n,ev,b in res])
def deg(cond, eps=1e-16):
    sk=np.sqrt(cond); r=(sk-1)/(sk+1); return np.ceil(np.log(eps)/np.log(r))+1
print('cond true mean/max', ct.mean(), ct.max(), 'bound', cb.mean(), cb.max())
print('deg true', deg(ct).mean(), deg(ct).max(), 'deg bound', deg(cb).mean(), deg(cb).max())
# deflating r top modes
for r in (4,8,16,32):
    cd = np.array([1+ev[r]/c for n,ev,b in res]); print('deflate',r,'cond',cd.mean(),cd.max(),'deg',deg(cd).mean(),deg(cd).max())
