"""Per-cycle convergence of the one-sided Jacobi on C2-like points with the kernel's ordering (odd-even transposition with
rotate-and-swap) and warm start from the level below: max |cos| and max |tan| met in each cycle.  numpy, CPU.
Usage: tools/sim_jacobi_cycles.py [WORKLOAD=C2-mini] [NCOL=6]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench_workload as bw
name = sys.argv[1] if len(sys.argv) > 1 else "C2-mini"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if name == "C2-sim":        # C2's 60 levels and lattice on a small horizontal grid (C2 itself: 14 GB of lists on the CPU)
    bw.CONFIGS["C2-sim"] = dict(bw.CONFIGS["C2"], nx=20, ny=20)
w = bw.build(name, torch.device('cpu'))
k, npts = w['k'], w['npts']
cfg = w['cfg']; nij = cfg['nx'] * cfg['ny']; nz = cfg['nz']
ens = w['ensval'][:, :k].numpy()

def amat(p):
    o0, o1 = int(w['obs_off'][p]), int(w['obs_off'][p + 1])
    idx = w['obs_idx'][o0:o1].numpy()
    wgt = 1.0 / w['rdiag'][o0:o1].numpy()
    Y = ens[idx]
    A = (Y * wgt[:, None]).T @ Y
    A[np.diag_indices(k)] += (k - 1.0)
    return A

def jacobi(G, maxcyc=12):
    """one-sided, columns of G; returns list of (maxcos, maxtan, n_violating_early) per cycle and final G"""
    n = G.shape[1]
    pos = list(range(n))          # column index at each line position
    out = []
    quiet2 = 0; stopped_at = None; pairs = 0
    for cyc in range(maxcyc):
        mc = mt = 0.0; viol = 0
        for step in range(n):
            start = step & 1
            for i in range(start, n - 1, 2):
                a_, b_ = pos[i], pos[i + 1]
                x, y = G[:, a_], G[:, b_]
                al, be, ga = x @ x, y @ y, x @ y
                cos = abs(ga) / np.sqrt(al * be)
                t = 0.0
                if cos > 1e-15:
                    h = 0.5 * (be - al)
                    t = ga * np.sign(h if h != 0 else 1.0) / (abs(h) + np.sqrt(h * h + ga * ga))
                    c = 1.0 / np.sqrt(1 + t * t)
                    G[:, a_], G[:, b_] = c * (x - t * y), c * (y + t * x)
                mc = max(mc, cos); mt = max(mt, abs(t))
                if cos > 1e-8 or abs(t) > 1e-6: viol += 1
                pos[i], pos[i + 1] = b_, a_
        out.append((mc, mt, viol))
        if viol == 0: break
    return out, G

rng = np.random.default_rng(1)
cols = rng.choice(nij, ncol, replace=False)
for col in cols:
    Q = None
    for lev in range(0, nz, max(1, nz // 6)):
        pass
    Q = None
    print("column", col); ncyc = []; lastviol = []
    for lev in range(nz):
        p = col + nij * lev
        A = amat(p)
        G0 = A if Q is None else A @ Q
        hist, G = jacobi(G0.copy())
        lam = np.sqrt((G * G).sum(0)); Q = G / lam
        ncyc.append(len(hist)); lastviol.append(hist[-2][2] if len(hist) > 1 else -1)
        if lev in (0, 1, 2, nz // 2, nz - 2):
            off0 = 0.0
            if lev:
                B = Qprev.T @ A @ Qprev
                off0 = np.sqrt(((B - np.diag(np.diag(B))) ** 2).sum()) / np.sqrt((B ** 2).sum())
            print("  lev", lev, "start off-norm %.1e" % off0, " per cycle (maxcos, maxtan, violations):",
                  " ".join("(%.0e,%.0e,%d)" % h for h in hist))
        Qprev = Q
    print("  cycles per level:", ncyc)
    print("  violations in the last non-quiet cycle:", lastviol)
