"""The eigen-free points of the staged path (k >= 63, loop body without k x k outputs;
csrc/letkf_staged.hip poly_apply, include/letkf_amd.h LETKF_OPT_STAGED_POLY): the transform, w-bar and the RTPS
quadratic form as Chebyshev expansions in M = Z Z^T + (k-1)/rho I instead of through mtx_eigen's replacement.
Each case runs the SAME call twice -- option on and off -- and requires (1) both within the loop body's tolerance of the
oracle's restatement of scale/letkf/letkf_tools.f90:313-527 (1e-10 * max(|x-bar|, |x'|) per variable, inflation 1e-12),
(2) the two within 1e-11 of each other (option off: the Jacobi eigen stage for k > 100, the two-wave Jacobi kernel of
csrc/letkf_wave.hip for 63 <= k <= 100), (3) the sweep counts to show which path a point took: nsweep = -(Chebyshev degree) with the option
on (2 <= n < k: expansion in the n x n matrix; n >= k, k <= 512: in the k x k matrix), > 0 with it off."""
import numpy as np
import pytest
import torch

import _oracle
from _cases import das_case

pytestmark = pytest.mark.gpu


def run(c, k, nv, cfg, poly):
    from _gpu import ctx, dev
    det = bool(cfg.get("det_run", 0))
    npts = c["npts"]
    anal = torch.full((c["gues"].size,), float("nan"), dtype=torch.float64, device="cuda")
    infl = dev(c["infl"])
    status = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    nsweep = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    cx = ctx()
    cx.set_option(cx.OPT_STAGED_POLY, 1 if poly else 0)
    try:
        cx.das_points(k, nv, dev(c["obs_off"]), dev(c["obs_idx"]), dev(c["rdiag"]), dev(c["rloc"]), dev(c["ensval"]),
                      c["kld"], dev(c["dep"]), infl, dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"],
                      beta=dev(c["beta"]), det_run=det, infl_adaptive=cfg.get("infl_adaptive", 0),
                      relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                      relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                      q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0),
                      iv_p=4, iv_q_first=5, iv_q_last=min(10, nv - 1), status=status, nsweep=nsweep)
        torch.cuda.synchronize()
    finally:
        cx.set_option(cx.OPT_STAGED_POLY, 1)
    return anal.cpu().numpy(), infl.cpu().numpy(), status.cpu().numpy(), nsweep.cpu().numpy()


def oracle(c, k, nv, cfg):
    det = bool(cfg.get("det_run", 0))
    prm = _oracle.DasParams(k=k, nv=nv, det_run=int(det), infl_adaptive=cfg.get("infl_adaptive", 0),
                            relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                            relax_alpha=cfg.get("relax_alpha", 0.0),
                            relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                            q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0),
                            iv_p=4, iv_q_first=5, iv_q_last=min(10, nv - 1), nthreads=4)
    ref = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], c["rdiag"], c["rloc"], c["ensval"], c["dep"],
                             c["beta"], c["infl"], c["gues"], c["sp"], c["sm"], c["sv"])
    assert ref["rc"] == 0
    return ref


@pytest.mark.parametrize("name", ["rtps_adaptive_det", "rtpp", "rtps_qtop", "norelax"])
@pytest.mark.parametrize("k,npts,nobs_tot,n_mean", [(64, 16, 300, 40), (80, 16, 400, 100), (100, 24, 500, 70), (100, 16, 600, 150), (144, 24, 500, 70), (144, 20, 600, 140), (320, 16, 700, 150), (320, 8, 900, 330), (1000, 4, 600, 200)])
def test_polynomial_points_equal_jacobi_points_and_oracle(name, k, npts, nobs_tot, n_mean):
    from test_gpu_das import CONFIGS, compare_anal
    cfg = CONFIGS[name]
    det = bool(cfg.get("det_run", 0))
    c = das_case(k=k, nv=11, npts=npts, nobs_tot=nobs_tot, n_mean=n_mean, seed=1000 + k, det_run=det, infl0=1.07)
    ref = oracle(c, k, 11, cfg)
    a1, i1, s1, w1 = run(c, k, 11, cfg, poly=True)
    a0, i0, s0, w0 = run(c, k, 11, cfg, poly=False)
    assert (s1 == 0).all() and (s0 == 0).all(), (s1, s0)
    compare_anal(c, ref, a1, k, 11, det)
    compare_anal(c, ref, a0, k, 11, det)
    assert np.abs(i1 - ref["infl"]).max() <= 1e-12 and np.abs(i0 - ref["infl"]).max() <= 1e-12
    nens = c["nens"]
    x = c["gues"].reshape(11, nens, npts)
    members = list(range(k)) + ([k + 1] if det else [])
    for v in range(11):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        d = np.abs(a1.reshape(11, nens, npts)[v, members] - a0.reshape(11, nens, npts)[v, members]).max()
        assert d <= 1e-11 * scale, (v, d, scale)
    n = np.diff(c["obs_off"])
    solved = (n > 0) & (c["beta"] != 0.0)
    small = solved & (n >= 2) & (n < k)
    assert small.sum() >= 2
    assert ((w1[small] < 0) & (w1[small] >= -128)).all(), (n[small], w1[small])   # took the eigen-free stage: -(iterations)
    assert (w0[small] > 0).all()                                    # ... and the Jacobi with the option off
    big = solved & (n >= k)
    # n >= k: the same expansion in the k x k matrix A = Z^T Z + c I (k <= 512 rows, degree <= 64), else the Jacobi
    assert (w1[big] != 0).all() and (w0[big] > 0).all()
    if k <= 512 and big.any():
        assert (w1[big] < 0).any(), (n[big], w1[big])


@pytest.mark.parametrize("scale", [1e-4, 1e-40])
def test_badly_conditioned_points_stay_on_the_jacobi(scale):
    """tiny observation errors: |Z Z^T| >> (k-1)/rho, the Chebyshev degree for 1e-16 exceeds the cap and the point keeps
    the eigen stage -- same answer either way"""
    from test_gpu_das import CONFIGS, compare_anal
    cfg = CONFIGS["rtps"]
    k = 144
    c = das_case(k=k, nv=11, npts=12, nobs_tot=400, n_mean=60, seed=77, infl0=1.0)
    c["rdiag"] = c["rdiag"] * scale                   # errors 100 x smaller; 1e-20 x: cond ~ 1e40, the degree estimate saturates
    a1, i1, s1, w1 = run(c, k, 11, cfg, poly=True)
    n = np.diff(c["obs_off"])
    if scale < 1e-10:                                 # (no parity claim at cond 1e40: only that the point is NOT taken)
        small = (n >= 2) & (c["beta"] != 0.0)
        assert small.any() and (w1[small] >= 0).all(), (n, w1)
        return
    ref = oracle(c, k, 11, cfg)
    assert (s1 == 0).all()
    compare_anal(c, ref, a1, k, 11, False)
    small = (n >= 8) & (n < k) & (c["beta"] != 0.0)
    assert small.any() and (w1[small] > 0).all(), (n, w1)
