"""Randomised soak of the batched fine boundary against the oracle: every ensemble size 2..62 (wave kernel) and a few
above (workgroup kernel), ragged nobsl incl. 0 / 1 / < k / >> k, random inflation, strongly varying obs errors."""
import numpy as np
import pytest
import torch

import _oracle
from _cases import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kset", [list(range(2, 22)), list(range(22, 42)), list(range(42, 63)), [63, 64, 65, 97]])
def test_soak(kset):
    from _gpu import ctx, dev
    rng = np.random.default_rng(sum(kset))
    worst = 0.0
    for k in kset:
        nobs, nb = 96, 6
        nobsl = rng.integers(0, nobs + 1, size=nb).astype(np.int32)
        nobsl[0], nobsl[1] = 0, min(nobs, max(1, k - 1))
        H = np.zeros((nb, k, nobs))
        rd = np.ones((nb, nobs)); rl = np.ones((nb, nobs)); dp = np.zeros((nb, nobs))
        infl = rng.uniform(0.9, 2.0, size=nb)
        exp = []
        for b in range(nb):
            n = int(nobsl[b])
            y = rng.standard_normal((nobs, k)) * rng.choice([0.1, 1.0, 10.0])
            y -= y.mean(axis=1, keepdims=True)
            err = rng.choice([0.05, 1.0, 30.0], size=nobs)
            rloc = np.exp(-0.5 * rng.uniform(0, 13.3, size=nobs))
            H[b] = y.T
            rd[b] = err ** 2 / rloc
            rl[b] = rloc
            dp[b] = rng.standard_normal(nobs) * err
            exp.append(_oracle.letkf_core("oracle", k, nobs, n, np.asfortranarray(y), rd[b], rl[b], dp[b],
                                          float(infl[b]), rdiag_wloc=True, infl_update=True))
        d_infl = dev(infl)
        trans = torch.zeros(nb, k * k, dtype=torch.float64, device="cuda")
        pao = torch.zeros_like(trans)
        transm = torch.zeros(nb, k, dtype=torch.float64, device="cuda")
        status = torch.full((nb,), -1, dtype=torch.int32, device="cuda")
        ctx().core_batch(k, nobs, dev(nobsl), dev(H), dev(rd), dev(rl), dev(dp), d_infl, trans, transm=transm, pao=pao,
                         rdiag_wloc=True, infl_update=True, status=status)
        torch.cuda.synchronize()
        st = status.cpu().numpy()
        assert set(st.tolist()) <= {0, 3}, (k, st)
        T, P, W, I = trans.cpu().numpy(), pao.cpu().numpy(), transm.cpu().numpy(), d_infl.cpu().numpy()
        for b in range(nb):
            e = exp[b]
            # cond(A) reaches ~1e5 here (obs error 0.05 against 30): allow cond * eps on top of the 1e-11 bar
            tol = 1e-11 if st[b] == 0 else 2e-9
            et = relerr(T[b].reshape(k, k).T, e["trans"])
            ep = relerr(P[b].reshape(k, k).T, e["pao"])
            worst = max(worst, et, ep)
            assert et <= max(tol, 1e-10) and ep <= max(tol, 1e-10), (k, b, int(nobsl[b]), et, ep)
            assert np.abs(W[b] - e["transm"]).max() <= 1e-9 * max(1.0, np.abs(e["transm"]).max())
            assert abs(I[b] - e["parm_infl"]) <= 1e-10
    print("worst relative error", worst)
