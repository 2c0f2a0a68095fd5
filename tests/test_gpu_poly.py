"""The eigen-free points of the staged path (k >= 63, loop body without k x k outputs; csrc/letkf_krylov.hip,
include/letkf_amd.h LETKF_OPT_STAGED_POLY): the transform, w-bar and the RTPS quadratic form of
scale/letkf/letkf_tools.f90:313-527 / common/common_letkf.f90:127-216 from conjugate gradients + the Lanczos tridiagonal
of M = Z Z^T + (k-1)/rho I (n < k) or A = Z^T Z + (k-1)/rho I (n >= k) instead of through mtx_eigen's replacement.

Each case runs the SAME call twice -- option on and off -- and requires (1) both within the loop body's tolerance of the
oracle's restatement (1e-10 * max(|x-bar|, |x'|) per variable, inflation 1e-12), (2) the two within 1e-11 of each other
(option off: the Jacobi eigen stage for k > 100, the two-wave Jacobi kernel of csrc/letkf_wave.hip for 63 <= k <= 100),
(3) the sweep counts to show which path a point took: nsweep = -(CG iterations) with the option on, > 0 with it off.

Round 3 adds what round 2's verdict found missing: spectra with a few dominant modes and obs-space spread of several
observation errors (cond 10 .. 1000: the Chebyshev route of round 2 fell back to the Jacobi above cond ~ 12), the stage's
own fall-back (no convergence within its 128 iterations -> eigen stage, inside the same call) at the orders production
uses (k = 64, 100 on the workgroup Jacobi; 320 dual), and the matrices it must not touch."""
import numpy as np
import pytest
import torch

import _oracle
from _cases import das_case

pytestmark = pytest.mark.gpu

MAX_ITER = 128   # letkf_krylov.hip kMmax


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


def point_conds(c, k):
    """cond(A) = 1 + lambda_max(Z^T Z) rho / (k - 1) per point (the shift uses the point's inflation slot)."""
    out = np.ones(c["npts"])
    for p in range(c["npts"]):
        o0, o1 = c["obs_off"][p], c["obs_off"][p + 1]
        if o1 - o0 < 1:
            continue
        z = c["ensval"][c["obs_idx"][o0:o1], :k] / np.sqrt(c["rdiag"][o0:o1])[:, None]
        s = np.linalg.svd(z, compute_uv=False)[0] ** 2
        out[p] = 1.0 + s * c["infl"][p] / (k - 1.0)
    return out


@pytest.mark.parametrize("name", ["rtps_adaptive_det", "rtpp", "rtps_qtop", "norelax"])
@pytest.mark.parametrize("k,npts,nobs_tot,n_mean", [(63, 12, 300, 40), (63, 12, 400, 90), (64, 16, 300, 40), (65, 12, 300, 80), (80, 16, 400, 100), (128, 12, 500, 60), (128, 12, 600, 170), (100, 24, 500, 70), (100, 16, 600, 150), (144, 24, 500, 70), (144, 20, 600, 140), (320, 16, 700, 150), (320, 8, 900, 330), (1000, 4, 600, 200)])
def test_eigenfree_points_equal_jacobi_points_and_oracle(name, k, npts, nobs_tot, n_mean):
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
    assert ((w1[small] < 0) & (w1[small] >= -MAX_ITER)).all(), (n[small], w1[small])   # took the eigen-free stage: -(iterations)
    assert (w0[small] > 0).all()                                    # ... and the Jacobi with the option off
    big = solved & (n >= k)
    # n >= k: the same iteration on the k x k matrix A = Z^T Z + c I (k <= 512 rows), else the Jacobi
    assert (w1[big] != 0).all() and (w0[big] > 0).all()
    if k <= 512 and big.any():
        assert (w1[big] < 0).all(), (n[big], w1[big])


def with_modes(c, k, spread, nmodes, seed):
    """Obs-space perturbations with `nmodes` dominant modes (every observation sees the same few member patterns with
    weights of one sign: what spatially correlated H(x) perturbations look like inside a localisation volume) on top of
    the independent part, scaled to a standard deviation of `spread` observation errors (mean error of the table 3)."""
    rng = np.random.default_rng(seed)
    nobs = c["ensval"].shape[0]
    y = c["ensval"][:, :k]
    for _ in range(nmodes):
        f = 0.5 + rng.random(nobs)
        m = rng.standard_normal(k)
        m -= m.mean()
        y += 1.5 * np.outer(f, m)
    y -= y.mean(axis=1, keepdims=True)
    y *= spread * 3.0 / y.std()
    c["ensval"][:, :k] = y


@pytest.mark.parametrize("k,npts,nobs_tot,n_mean,spread", [
    (64, 12, 400, 100, 2.0), (100, 16, 600, 150, 1.6), (100, 16, 600, 150, 2.4), (100, 12, 600, 60, 4.0),
    (128, 12, 600, 100, 2.4), (320, 12, 900, 200, 1.6), (320, 12, 900, 200, 2.4), (320, 8, 900, 200, 6.0),
    (1000, 4, 600, 200, 2.4)])
def test_dominant_modes_and_large_obs_space_spread(k, npts, nobs_tot, n_mean, spread):
    """What carried round 2's headline only on benign spectra: cond(A) of 10 .. 1000 with a few dominant modes.  Every
    solved point stays on the eigen-free stage (no fall-back), within its iteration budget, at the loop body's tolerance."""
    from test_gpu_das import CONFIGS, compare_anal
    cfg = CONFIGS["rtps_adaptive_det"]
    c = das_case(k=k, nv=11, npts=npts, nobs_tot=nobs_tot, n_mean=n_mean, seed=4000 + k, det_run=True, infl0=1.03, vary_n=False)
    with_modes(c, k, spread, 3, seed=k)
    cond = point_conds(c, k)
    assert cond.max() > 10.0, cond
    ref = oracle(c, k, 11, cfg)
    a1, i1, s1, w1 = run(c, k, 11, cfg, poly=True)
    assert (s1 == 0).all(), s1
    compare_anal(c, ref, a1, k, 11, True)
    assert np.abs(i1 - ref["infl"]).max() <= 1e-12
    solved = c["beta"] != 0.0
    assert ((w1[solved] < 0) & (w1[solved] >= -MAX_ITER)).all(), (cond[solved], w1[solved])
    # the iteration count follows the spectrum's shape, not its width: far below the ~ 18.7 sqrt(cond) terms a polynomial on
    # the whole interval needs
    if cond.max() > 30.0:
        assert (-w1[solved]).max() < 10.0 * np.sqrt(cond.max()), (cond.max(), w1)


@pytest.mark.parametrize("k,n_mean", [(64, 60), (100, 96), (100, 140), (320, 180), (320, 260), (250, 400)])
def test_points_the_iteration_gives_up_go_to_the_eigen_stage(k, n_mean):
    """Observation errors 100 x smaller: a flat spectrum of width cond ~ 1e4 .. 1e5 at order min(n, k) -- conjugate
    gradients cannot finish within 128 iterations where the order is well above that, the stage rewrites the point's
    solver and the eigen stage of the SAME call analyses it (workgroup Jacobi at orders <= 208, block Jacobi above):
    status 0, result within cond * eps of the oracle."""
    from test_gpu_das import CONFIGS
    cfg = CONFIGS["rtps"]
    c = das_case(k=k, nv=11, npts=10, nobs_tot=600, n_mean=n_mean, seed=5000 + k, infl0=1.0, vary_n=False)
    c["rdiag"] = c["rdiag"] * 1e-4
    cond = point_conds(c, k)
    ref = oracle(c, k, 11, cfg)
    a1, i1, s1, w1 = run(c, k, 11, cfg, poly=True)
    a0, i0, s0, w0 = run(c, k, 11, cfg, poly=False)
    assert (s1 == 0).all() and (s0 == 0).all(), (s1, s0)
    solved = c["beta"] != 0.0
    assert (w1[solved] != 0).all()
    if min(k, n_mean) > MAX_ITER:
        assert (w1[solved] > 0).all(), w1          # gave up -> Jacobi sweeps
    nens, npts = c["nens"], c["npts"]
    x = c["gues"].reshape(11, nens, npts)
    tol = max(1e-10, 4.0 * cond.max() * 2.2e-16)   # both EISPACK's QL and the Jacobi are accurate to cond * eps only
    for a in (a1, a0):
        for v in range(11):
            scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
            err = np.abs(a.reshape(11, nens, npts)[v, :k] - ref["anal"].reshape(11, nens, npts)[v, :k]).max()
            assert err <= tol * scale, (v, err, scale, cond.max())


@pytest.mark.parametrize("k,n_mean,scale", [(320, 260, 1e-2), (320, 260, 1e-6), (512, 300, 1e-3), (512, 300, 1e-5), (250, 400, 1e-2), (250, 400, 1e-6)])
def test_block_jacobi_orders_hold_to_cond_1e6(k, n_mean, scale):
    """Orders above 208 on the eigen stage (LETKF_OPT_STAGED_POLY = 0: every point) are the block Jacobi on the matrix cores,
    its 32-column problems solved through their Cholesky factor, finished by the scalar one-sided iteration once the block
    sweeps stagnate (letkf_kernels.hip, jacobi_block_mfma).  Round 2's version reported status 1 from cond(A) ~ 1e3 on
    (tools/r3_probe_block_jacobi.py); the reference's tred2 / tql2 has no such limit.  Status 0 and cond * eps accuracy."""
    from test_gpu_das import CONFIGS
    cfg = CONFIGS["rtps"]
    c = das_case(k=k, nv=11, npts=6, nobs_tot=600, n_mean=n_mean, seed=5000 + k, infl0=1.0, vary_n=False)
    c["rdiag"] = c["rdiag"] * scale
    cond = point_conds(c, k)
    ref = oracle(c, k, 11, cfg)
    a0, i0, s0, w0 = run(c, k, 11, cfg, poly=False)
    assert (s0 == 0).all(), (s0, w0)
    solved = c["beta"] != 0.0
    assert (w0[solved] > 0).all() and (w0[solved] < 30).all(), w0
    nens, npts = c["nens"], c["npts"]
    x = c["gues"].reshape(11, nens, npts)
    tol = max(1e-12, 8.0 * cond.max() * 2.2e-16)
    for v in range(11):
        sc = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        err = np.abs(a0.reshape(11, nens, npts)[v, :k] - ref["anal"].reshape(11, nens, npts)[v, :k]).max()
        assert err <= tol * sc, (v, err, sc, cond.max())


def test_hopeless_matrices_are_not_iterated_on():
    """Observation errors 1e-20 x: the mean eigenvalue of Z Z^T is beyond 1e5 (k-1)/rho, no iteration is attempted (the
    Gram stage keeps the point on the eigen stage; no parity claim at cond 1e40, only that the point is NOT taken)."""
    from test_gpu_das import CONFIGS
    cfg = CONFIGS["rtps"]
    k = 144
    c = das_case(k=k, nv=11, npts=12, nobs_tot=400, n_mean=60, seed=77, infl0=1.0)
    c["rdiag"] = c["rdiag"] * 1e-40
    a1, i1, s1, w1 = run(c, k, 11, cfg, poly=True)
    n = np.diff(c["obs_off"])
    small = (n >= 2) & (c["beta"] != 0.0)
    assert small.any() and (w1[small] >= 0).all(), (n, w1)


def test_nan_in_the_observation_table_is_reported_not_iterated_away():
    """A NaN in a local observation's row: the iteration sees NaN sums, hands the point to the eigen stage, which reports
    it (status 1: not converged) -- the other points of the batch are unaffected."""
    from test_gpu_das import CONFIGS, compare_anal
    cfg = CONFIGS["rtps"]
    k = 100
    c = das_case(k=k, nv=11, npts=12, nobs_tot=500, n_mean=80, seed=99, infl0=1.0, vary_n=False)
    ref = oracle(c, k, 11, cfg)
    bad_row = int(c["obs_idx"][c["obs_off"][3]])
    c["ensval"] = c["ensval"].copy()
    c["ensval"][bad_row, 5] = np.nan
    a1, i1, s1, w1 = run(c, k, 11, cfg, poly=True)
    touched = np.array([bad_row in c["obs_idx"][c["obs_off"][p]:c["obs_off"][p + 1]] for p in range(c["npts"])])
    solved = c["beta"] != 0.0
    assert (s1[touched & solved] != 0).all(), s1
    assert (s1[~touched] == 0).all(), s1
    nens, npts = c["nens"], c["npts"]
    x = c["gues"].reshape(11, nens, npts)
    for v in range(11):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        err = np.abs(a1.reshape(11, nens, npts)[v, :k][:, ~touched] - ref["anal"].reshape(11, nens, npts)[v, :k][:, ~touched]).max()
        assert err <= 1e-10 * scale


@pytest.mark.parametrize("k,nv,n_mean", [(8, 3, 5), (8, 14, 30), (20, 7, 12), (33, 5, 50), (50, 13, 120), (62, 1, 40)])
@pytest.mark.parametrize("name", ["rtps_adaptive_det", "rtpp"])
def test_small_ensembles_with_another_number_of_variables(name, k, nv, n_mean):
    """nv != 11 does not have a register-kernel instantiation: such calls take the staged path at any k, i.e. the eigen-free
    stage on matrices of order 2 .. 62 (one 16-row block per wave, most waves idle) -- tiny orders, n = 1 points, ragged n."""
    from test_gpu_das import CONFIGS
    cfg = dict(CONFIGS[name])
    det = bool(cfg.get("det_run", 0))
    c = das_case(k=k, nv=nv, npts=40, nobs_tot=300, n_mean=n_mean, seed=6000 + 10 * k + nv, det_run=det, infl0=1.05)
    ref = oracle(c, k, nv, cfg)
    a1, i1, s1, w1 = run(c, k, nv, cfg, poly=True)
    assert (s1 == 0).all(), s1
    nens, npts = c["nens"], c["npts"]
    x = c["gues"].reshape(nv, nens, npts)
    members = list(range(k)) + ([k + 1] if det else [])
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        err = np.abs(a1.reshape(nv, nens, npts)[v, members] - ref["anal"].reshape(nv, nens, npts)[v, members]).max()
        assert err <= 1e-10 * scale, (v, err, scale)
    assert np.abs(i1 - ref["infl"]).max() <= 1e-12
    n = np.diff(c["obs_off"])
    solved = (n >= 2) & (c["beta"] != 0.0)
    assert solved.any() and (w1[solved] < 0).all(), (n[solved], w1[solved])
