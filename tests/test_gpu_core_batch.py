"""GPU parity, batched fine boundary (letkf_core_batch_dev) against the oracle on seeded inputs: ragged nobsl
(including 0 and nobs), both rdiag_wloc modes, optional outputs present/absent, adaptive inflation."""
import numpy as np
import pytest
import torch

import _oracle
from _cases import core_case, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-11


@pytest.mark.parametrize("k,nobs,nb", [(20, 64, 24), (50, 256, 40), (64, 100, 16), (33, 70, 12), (100, 120, 6)])
@pytest.mark.parametrize("wloc,iu,det", [(True, True, True), (False, False, False)])
def test_core_batch(k, nobs, nb, wloc, iu, det):
    from _gpu import ctx, dev
    rng = np.random.default_rng(k * 7 + nobs)
    nobsl = rng.integers(0, nobs + 1, size=nb).astype(np.int32)
    nobsl[0] = 0
    nobsl[1] = nobs
    nobsl[2] = 1
    H = np.zeros((nb, k, nobs))
    rd = np.zeros((nb, nobs)); rl = np.zeros((nb, nobs)); dp = np.zeros((nb, nobs)); dd = np.zeros((nb, nobs))
    infl = rng.uniform(1.0, 1.3, size=nb)
    exp = []
    for b in range(nb):
        c = core_case(k, int(nobsl[b]), seed=1000 + b, nobs=nobs, rdiag_wloc=wloc, infl=float(infl[b]), with_det=True)
        H[b] = c["hdxb"].T  # (k, nobs) C-order == column-major (nobs, k)
        rd[b], rl[b], dp[b], dd[b] = c["rdiag"], c["rloc"], c["dep"], c["depd"]
        exp.append(_oracle.letkf_core("oracle", k, nobs, int(nobsl[b]), c["hdxb"], c["rdiag"], c["rloc"], c["dep"],
                                      float(infl[b]), rdiag_wloc=wloc, infl_update=iu,
                                      depd=c["depd"] if det else None, want_transmd=det))
    d_infl = dev(infl)
    trans = torch.zeros(nb, k * k, dtype=torch.float64, device="cuda")
    pao = torch.zeros_like(trans)
    transm = torch.zeros(nb, k, dtype=torch.float64, device="cuda")
    transmd = torch.zeros_like(transm)
    status = torch.full((nb,), -1, dtype=torch.int32, device="cuda")
    nsweep = torch.zeros(nb, dtype=torch.int32, device="cuda")
    ctx().core_batch(k, nobs, dev(nobsl), dev(H), dev(rd), dev(rl), dev(dp), d_infl, trans, transm=transm, pao=pao,
                     depd=dev(dd) if det else None, transmd=transmd if det else None, rdiag_wloc=wloc,
                     infl_update=iu, status=status, nsweep=nsweep)
    torch.cuda.synchronize()
    assert status.cpu().numpy().tolist() == [0] * nb
    T = trans.cpu().numpy(); P = pao.cpu().numpy(); W = transm.cpu().numpy(); WD = transmd.cpu().numpy()
    I = d_infl.cpu().numpy()
    for b in range(nb):
        e = exp[b]
        assert relerr(T[b].reshape(k, k).T, e["trans"]) <= TOL, (b, nobsl[b])
        assert relerr(P[b].reshape(k, k).T, e["pao"]) <= TOL, (b, nobsl[b])
        assert np.abs(W[b] - e["transm"]).max() <= TOL * max(1.0, np.abs(e["transm"]).max())
        if det:
            assert np.abs(WD[b] - e["transmd"]).max() <= TOL * max(1.0, np.abs(e["transmd"]).max())
        assert abs(I[b] - e["parm_infl"]) <= 1e-12
    assert int(nsweep.max()) < 30


@pytest.mark.parametrize("k", [2, 3, 15, 16, 17, 21, 32, 33, 47, 48, 49, 51, 62, 63, 65, 80, 81, 99])
def test_core_batch_ensemble_sizes_at_the_instantiation_bounds(k):
    """every KR instantiation of the register kernel at its smallest and largest ensemble size, T / Pa / w-bar out (the KKOUT
    twins: r4 found a hipcc defect in two of them, DESIGN.md section 8 -- sizes the parametrisation above never launched)"""
    test_core_batch(k, 44, 7, True, True, True)


def test_core_batch_without_transm_folds_wbar():
    """transm absent -> w-bar is added to every column of trans (common/common_letkf.f90:218-226)."""
    from _gpu import ctx, dev
    k, nobs, nb = 20, 40, 3
    H = np.zeros((nb, k, nobs)); rd = np.zeros((nb, nobs)); rl = np.zeros((nb, nobs)); dp = np.zeros((nb, nobs))
    exp = []
    for b in range(nb):
        c = core_case(k, nobs, seed=77 + b, nobs=nobs)
        H[b] = c["hdxb"].T
        rd[b], rl[b], dp[b] = c["rdiag"], c["rloc"], c["dep"]
        exp.append(_oracle.letkf_core("oracle", k, nobs, nobs, c["hdxb"], c["rdiag"], c["rloc"], c["dep"], 1.0,
                                      want_transm=False, want_pao=False, rdiag_wloc=True))
    trans = torch.zeros(nb, k * k, dtype=torch.float64, device="cuda")
    ctx().core_batch(k, nobs, dev(np.full(nb, nobs, dtype=np.int32)), dev(H), dev(rd), dev(rl), dev(dp),
                     dev(np.ones(nb)), trans, rdiag_wloc=True)
    torch.cuda.synchronize()
    for b in range(nb):
        assert relerr(trans[b].cpu().numpy().reshape(k, k).T, exp[b]["trans"]) <= TOL
