"""GPU parity, coarse boundary: letkf_das_points_dev (the das_letkf loop body, scale/letkf/letkf_tools.f90:313-527)
against the oracle's restatement on the same seeded inputs.  Tolerance:
|d xa| <= 1e-10 * max(|x-bar|, |x'|) per variable; adaptive inflation 1e-12."""
import numpy as np
import pytest
import torch

import _oracle
from _cases import das_case

pytestmark = pytest.mark.gpu

CONFIGS = {
    "rtps": dict(relax_alpha_spread=0.95),
    "rtpp": dict(relax_alpha=0.7, relax_to_inflated_prior=1),
    "norelax": dict(),
    "rtps_det": dict(relax_alpha_spread=0.95, det_run=1),
    "rtps_adaptive_det": dict(relax_alpha_spread=0.8, det_run=1, infl_adaptive=1, relax_to_inflated_prior=1),
    "rtpp_qclamp": dict(relax_alpha=0.5, q_sprd_max=0.05),
    "rtps_qtop": dict(relax_alpha_spread=0.95, q_update_top=49.0, det_run=1, infl_adaptive=1),
}


def run_both(k, nv, npts, nobs_tot, n_mean, seed, cfg, want_trans=False, warm_run=0, warm_stride=0):
    from _gpu import ctx, dev
    det = bool(cfg.get("det_run", 0))
    c = das_case(k=k, nv=nv, npts=npts, nobs_tot=nobs_tot, n_mean=n_mean, seed=seed, det_run=det, infl0=1.07)
    prm = _oracle.DasParams(k=k, nv=nv, det_run=int(det), infl_adaptive=cfg.get("infl_adaptive", 0),
                            relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                            relax_alpha=cfg.get("relax_alpha", 0.0),
                            relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                            q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0),
                            iv_p=4, iv_q_first=5, iv_q_last=min(10, nv - 1), nthreads=4)
    ref = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], c["rdiag"], c["rloc"], c["ensval"], c["dep"],
                             c["beta"], c["infl"], c["gues"], c["sp"], c["sm"], c["sv"], want_trans=want_trans,
                             want_pa=want_trans and cfg.get("relax_alpha_spread", 0.0) != 0.0)
    assert ref["rc"] == 0
    anal = torch.full((c["gues"].size,), float("nan"), dtype=torch.float64, device="cuda")
    infl = dev(c["infl"])
    status = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    trans = torch.zeros(npts, k * k, dtype=torch.float64, device="cuda") if want_trans else None
    transm = torch.zeros(npts, k, dtype=torch.float64, device="cuda") if want_trans else None
    ctx().das_points(k, nv, dev(c["obs_off"]), dev(c["obs_idx"]), dev(c["rdiag"]), dev(c["rloc"]), dev(c["ensval"]),
                     c["kld"], dev(c["dep"]), infl, dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"],
                     beta=dev(c["beta"]), det_run=det, infl_adaptive=cfg.get("infl_adaptive", 0),
                     relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                     relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                     q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0),
                     iv_p=4, iv_q_first=5, iv_q_last=min(10, nv - 1), trans_out=trans, transm_out=transm,
                     status=status, warm_run=warm_run, warm_stride=warm_stride)
    torch.cuda.synchronize()
    return c, ref, anal.cpu().numpy(), infl.cpu().numpy(), status.cpu().numpy(), trans, transm


def compare_anal(c, ref, got, k, nv, det):
    nens, npts = c["nens"], c["npts"]
    g = got.reshape(nv, nens, npts)
    e = ref["anal"].reshape(nv, nens, npts)
    x = c["gues"].reshape(nv, nens, npts)
    members = list(range(k)) + ([k + 1] if det else [])
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        err = np.abs(g[v, members] - e[v, members]).max()
        assert np.isfinite(g[v, members]).all()
        assert err <= 1e-10 * scale, (v, err, scale)
    # slot k (the mean) is not written by the path: the reference fills it later with ensmean_grd (letkf.f90:207)
    assert np.isnan(g[:, k]).all()


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("k,npts,nobs_tot,n_mean", [(20, 48, 400, 60), (50, 40, 900, 200)])
def test_das_points_vs_oracle(name, k, npts, nobs_tot, n_mean):
    cfg = CONFIGS[name]
    c, ref, got, infl, status, _, _ = run_both(k, 11, npts, nobs_tot, n_mean, seed=31 + k, cfg=cfg)
    assert (status == 0).all(), status
    compare_anal(c, ref, got, k, 11, bool(cfg.get("det_run", 0)))
    assert np.abs(infl - ref["infl"]).max() <= 1e-12


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("k,npts,nobs_tot,n_mean,warm_run", [(20, 150, 500, 70, 0), (19, 131, 500, 60, 7), (16, 100, 400, 50, 5), (9, 77, 300, 30, 4), (2, 50, 100, 9, 3)])
def test_das_points_small_ensembles_on_both_kernels(name, k, npts, nobs_tot, n_mean, warm_run):
    """k <= 20: three points per wave (letkf_trio.hip, the default: one eigensolve for three neighbouring runs, warm starts from
    the park; k = 19, 20 fill all 32 slots of the line, k = 9 and 2 a few) and the one-point register kernel
    (LETKF_OPT_SMALL_K_TRIO = 0), both against the oracle, every switch; runs of several lengths, batches that are not a multiple
    of three runs, points without observations and beta = 0 points inside the runs."""
    from _gpu import ctx
    cfg = CONFIGS[name]
    for trio in (1, 0):
        ctx().set_option(ctx().OPT_SMALL_K_TRIO, trio)
        try:
            c, ref, got, infl, status, _, _ = run_both(k, 11, npts, nobs_tot, n_mean, seed=500 + k, cfg=cfg, warm_run=warm_run)
            assert ctx().last_path().startswith("letkf_trio_kernel" if trio else "letkf_wave_kernel"), ctx().last_path()
        finally:
            ctx().set_option(ctx().OPT_SMALL_K_TRIO, 1)
        assert (status == 0).all(), status
        compare_anal(c, ref, got, k, 11, bool(cfg.get("det_run", 0)))
        assert np.abs(infl - ref["infl"]).max() <= 1e-12


def test_das_points_weights_match():
    """the optional per-point trans / transm outputs equal what letkf_core returns inside the oracle"""
    cfg = CONFIGS["rtps"]
    k = 50
    c, ref, got, infl, status, trans, transm = run_both(k, 11, 24, 600, 150, seed=5, cfg=cfg, want_trans=True)
    T = trans.cpu().numpy()
    W = transm.cpu().numpy()
    for p in range(24):
        if c["beta"][p] == 0.0:
            continue
        den = np.abs(ref["trans"][p]).max()
        assert np.abs(T[p] - ref["trans"][p]).max() <= 1e-11 * den
        assert np.abs(W[p] - ref["transm"][p]).max() <= 1e-11 * max(1.0, np.abs(ref["transm"][p]).max())


@pytest.mark.parametrize("k", [50, 100])
def test_rtps_factor_diagnostic(k):
    """work3da of RELAX_SPREAD_OUT (scale/letkf/letkf_tools.f90:271-276, 460-462): the RTPS factor per variable"""
    from _gpu import ctx, dev
    cfg = dict(relax_alpha_spread=0.9, q_update_top=49.0, det_run=1)
    c = das_case(k=k, nv=11, npts=24, nobs_tot=500, n_mean=120, seed=77, det_run=True, infl0=1.02)
    prm = _oracle.DasParams(k=k, nv=11, det_run=1, infl_adaptive=0, relax_to_inflated_prior=1, relax_alpha=0.0,
                            relax_alpha_spread=0.9, q_update_top=49.0, q_sprd_max=0.0, iv_p=4, iv_q_first=5,
                            iv_q_last=10, nthreads=2)
    ref = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], c["rdiag"], c["rloc"], c["ensval"], c["dep"],
                             c["beta"], c["infl"], c["gues"], c["sp"], c["sm"], c["sv"], want_rtps=True)
    anal = torch.zeros(c["gues"].size, dtype=torch.float64, device="cuda")
    rt = torch.full((24 * 11,), -7.0, dtype=torch.float64, device="cuda")
    ctx().das_points(k, 11, dev(c["obs_off"]), dev(c["obs_idx"]), dev(c["rdiag"]), dev(c["rloc"]), dev(c["ensval"]),
                     c["kld"], dev(c["dep"]), dev(c["infl"]), dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"],
                     beta=dev(c["beta"]), det_run=True, relax_to_inflated_prior=1, relax_alpha_spread=0.9,
                     q_update_top=49.0, rtps_infl_out=rt)
    torch.cuda.synchronize()
    got = rt.cpu().numpy()
    assert np.abs(got - ref["rtps"]).max() <= 1e-11 * np.abs(ref["rtps"]).max()
    assert (ref["rtps"] != 1.0).sum() > 50


def test_das_points_odd_k_and_other_nv():
    c, ref, got, infl, status, _, _ = run_both(33, 5, 20, 300, 50, seed=9, cfg=dict(relax_alpha_spread=0.9))
    assert (status == 0).all()
    compare_anal(c, ref, got, 33, 5, False)


def test_das_points_k100():
    c, ref, got, infl, status, _, _ = run_both(100, 11, 12, 700, 150, seed=3, cfg=dict(relax_alpha_spread=0.95))
    assert (status == 0).all()
    compare_anal(c, ref, got, 100, 11, False)


def test_empty_batch_and_all_beta_zero():
    from _gpu import ctx, dev
    k, nv = 20, 11
    c = das_case(k=k, nv=nv, npts=8, nobs_tot=50, n_mean=10, seed=1)
    c["beta"][:] = 0.0
    anal = torch.zeros(c["gues"].size, dtype=torch.float64, device="cuda")
    ctx().das_points(k, nv, dev(c["obs_off"]), dev(c["obs_idx"]), dev(c["rdiag"]), dev(c["rloc"]), dev(c["ensval"]),
                     c["kld"], dev(c["dep"]), dev(c["infl"]), dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"],
                     beta=dev(c["beta"]))
    torch.cuda.synchronize()
    x = c["gues"].reshape(nv, c["nens"], 8)
    a = anal.cpu().numpy().reshape(nv, c["nens"], 8)
    assert np.array_equal(a[:, :k], x[:, :k] + x[:, k:k + 1])   # letkf_tools.f90:333-341, bit exact


@pytest.mark.parametrize("name", ["rtps", "rtps_adaptive_det", "rtps_qtop", "rtpp"])
@pytest.mark.parametrize("k,warm_run", [(50, 5), (20, 16), (33, 3), (62, 4), (100, 3), (72, 4), (65, 2), (63, 3), (64, 5)])
def test_das_points_warm_started_runs(name, k, warm_run):
    """Eigensolves warm-started from the previous point of a run (letkf_das_args.warm_run): same parity bar as the cold
    start, here on UNRELATED consecutive points (the worst case for the predictor), incl. points without obs and
    beta = 0 points inside a run, and T itself at 1e-11."""
    cfg = CONFIGS[name]
    npts = 43
    c, ref, got, infl, status, trans, transm = run_both(k, 11, npts, 600, 120, seed=100 + k, cfg=cfg, want_trans=True,
                                                        warm_run=warm_run)
    assert (status == 0).all()
    compare_anal(c, ref, got, k, 11, bool(cfg.get("det_run", 0)))
    assert np.abs(infl - ref["infl"]).max() <= 1e-12
    t = trans.cpu().numpy()
    for p in range(npts):
        if c["beta"][p] == 0.0:
            continue
        assert np.abs(t[p] - ref["trans"][p]).max() <= 1e-11 * np.abs(ref["trans"][p]).max(), p


@pytest.mark.parametrize("name", ["rtps_adaptive_det", "rtpp"])
@pytest.mark.parametrize("k,warm_run,stride", [(50, 0, 7), (50, 4, 6), (20, 3, 14), (33, 0, 42), (62, 2, 21), (100, 0, 6), (64, 5, 3),
                                               (50, 7, 1)])
def test_das_points_strided_warm_runs(name, k, warm_run, stride):
    """letkf_das_args.warm_stride: the runs walk points p, p + S, ... (up a column when S = nij1) -- every point solved
    exactly once, same parity bar, whatever the run length (0 = whole column; lengths that do not divide the column)."""
    cfg = CONFIGS[name]
    npts = 42
    c, ref, got, infl, status, trans, transm = run_both(k, 11, npts, 600, 120, seed=300 + k + stride, cfg=cfg,
                                                        want_trans=True, warm_run=warm_run, warm_stride=stride)
    assert (status == 0).all()
    compare_anal(c, ref, got, k, 11, bool(cfg.get("det_run", 0)))
    assert np.abs(infl - ref["infl"]).max() <= 1e-12
    t = trans.cpu().numpy()
    for p in range(npts):
        if c["beta"][p] == 0.0:
            continue
        assert np.abs(t[p] - ref["trans"][p]).max() <= 1e-11 * np.abs(ref["trans"][p]).max(), p


def test_warm_stride_must_divide_npts():
    with pytest.raises(RuntimeError, match="warm_stride"):
        run_both(50, 11, 42, 600, 120, seed=1, cfg=CONFIGS["rtpp"], warm_stride=5)


def test_warm_start_cuts_sweeps_on_neighbouring_points():
    """On a real grid (C2-mini: ij-fastest order, neighbours share most of their obs) the warm start must give the same
    analysis as the cold start and need clearly fewer Jacobi sweeps (simulation: 9.1 -> 6.0 for x-neighbours)."""
    import bench_workload as bw
    from _gpu import ctx
    w = bw.build("C2-mini", torch.device("cuda"))
    k, nv, npts = w["k"], w["nv"], w["npts"]
    res = {}
    for warm in (1, 16):
        anal = torch.zeros_like(w["gues"])
        infl = torch.ones(npts * nv, dtype=torch.float64, device="cuda")
        ns = torch.zeros(npts, dtype=torch.int32, device="cuda")
        st = torch.zeros(npts, dtype=torch.int32, device="cuda")
        ctx().das_points(k, nv, w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"], w["ensval"], w["kld"], w["dep"],
                         infl, w["gues"], anal, w["sp"], w["sm"], w["sv"], relax_alpha_spread=0.95, nsweep=ns,
                         status=st, warm_run=warm)
        torch.cuda.synchronize()
        assert int(st.abs().max()) == 0
        res[warm] = (anal.view(nv, w["nens"], npts)[:, :k].clone(), ns.double().mean().item())
    a_cold, s_cold = res[1]
    a_warm, s_warm = res[16]
    x = w["gues"].view(nv, w["nens"], npts)
    for v in range(nv):
        scale = max(x[v, k].abs().max().item(), x[v, :k].abs().max().item())
        assert (a_cold[v] - a_warm[v]).abs().max().item() <= 1e-10 * scale
    assert s_warm < s_cold - 1.5, (s_warm, s_cold)


@pytest.mark.parametrize("k,nv", [(50, 11), (20, 11), (100, 11), (33, 7)])
@pytest.mark.parametrize("name", ["rtps_adaptive_det", "rtps_qtop", "rtpp_qclamp"])
def test_two_variable_localisation_classes(name, k, nv):
    """var_local_n2nc_max = 2 (letkf_tools.f90:130-157, :372-439): one call per class with the class's own local lists
    (here the second class sees the same observations with VAR_LOCAL = 0.6: rloc * 0.6, rdiag / 0.6) and var_mask;
    every variable, inflation slot and RTPS factor must come out as the oracle's two-pass restatement."""
    from _gpu import ctx, dev
    cfg = CONFIGS[name]
    det = bool(cfg.get("det_run", 0))
    npts = 30
    c = das_case(k=k, nv=nv, npts=npts, nobs_tot=500, n_mean=90, seed=40 + k, det_run=det, infl0=1.07)
    c["infl"] = c["infl"] * (1.0 + 0.01 * np.arange(c["infl"].size) / c["infl"].size)   # distinct slots
    mask_a = 0b0011111 if nv == 7 else 0b00000011111
    mask_b = ((1 << nv) - 1) & ~mask_a
    lists = {mask_a: (c["rdiag"], c["rloc"]), mask_b: (c["rdiag"] / 0.6, c["rloc"] * 0.6)}
    want_anal = np.full_like(c["gues"], np.nan)
    want_infl = c["infl"].copy()
    want_rtps = np.full(npts * nv, np.nan)
    for mask, (rd, rl) in lists.items():
        prm = _oracle.DasParams(k=k, nv=nv, det_run=int(det), infl_adaptive=cfg.get("infl_adaptive", 0),
                                relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                                relax_alpha=cfg.get("relax_alpha", 0.0),
                                relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                                q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0),
                                iv_p=4, iv_q_first=5, iv_q_last=min(10, nv - 1), nthreads=4, var_mask=mask)
        r = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], rd, rl, c["ensval"], c["dep"], c["beta"], want_infl,
                               c["gues"], c["sp"], c["sm"], c["sv"], want_rtps=True)
        assert r["rc"] == 0
        sel = np.array([(mask >> v) & 1 for v in range(nv)], bool)
        wa, ra = want_anal.reshape(nv, -1), r["anal"].reshape(nv, -1)
        wa[sel] = ra[sel]
        want_infl = r["infl"]
        want_rtps.reshape(nv, npts)[sel] = r["rtps"].reshape(nv, npts)[sel]
    anal = torch.full((c["gues"].size,), float("nan"), dtype=torch.float64, device="cuda")
    infl = dev(c["infl"])
    rtps = torch.full((npts * nv,), float("nan"), dtype=torch.float64, device="cuda")
    status = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    for mask, (rd, rl) in lists.items():
        ctx().das_points(k, nv, dev(c["obs_off"]), dev(c["obs_idx"]), dev(rd), dev(rl), dev(c["ensval"]), c["kld"],
                         dev(c["dep"]), infl, dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"], beta=dev(c["beta"]),
                         det_run=det, infl_adaptive=cfg.get("infl_adaptive", 0),
                         relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                         relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                         q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0), iv_p=4,
                         iv_q_first=5, iv_q_last=min(10, nv - 1), status=status, rtps_infl_out=rtps, var_mask=mask)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all()
    compare_anal(c, dict(anal=want_anal), anal.cpu().numpy(), k, nv, det)
    assert np.abs(infl.cpu().numpy() - want_infl).max() <= 1e-12
    if cfg.get("relax_alpha_spread", 0.0) != 0.0:
        g = rtps.cpu().numpy()
        live = ~np.isnan(want_rtps) & (want_rtps != 0.0)
        assert np.allclose(g[live], want_rtps[live], rtol=1e-9, atol=0)


@pytest.mark.parametrize("k,name", [(144, "rtps_adaptive_det"), (160, "rtpp_qclamp")])
def test_das_points_large_k_block_jacobi(k, name):
    """k > 128: the workgroup kernel with G in the HBM workspace and the block Jacobi on the matrix cores"""
    cfg = CONFIGS[name]
    c, ref, got, infl, status, _, _ = run_both(k, 11, 7, 700, 180, seed=31 + k, cfg=cfg)
    assert (status == 0).all()
    compare_anal(c, ref, got, k, 11, bool(cfg.get("det_run", 0)))
    assert np.abs(infl - ref["infl"]).max() <= 1e-12


@pytest.mark.parametrize("k,nv,name", [(50, 16, "rtps_det"), (20, 15, "rtpp"), (120, 16, "rtps_adaptive_det"), (50, 7, "rtps"), (100, 5, "rtps_det")])
def test_das_points_other_variable_counts(k, nv, name):
    """nv != 11: the register kernels are instantiated for the reference's nv3d = 11 (common_nml.f90:19); other counts go to the
    staged path (nv + 2 <= 16 right-hand sides) and, beyond that, to round 1's workgroup kernel (letkf_point_kernel) -- the one
    production route that kernel still has.  Against the oracle like every other route."""
    from _gpu import ctx
    cfg = CONFIGS[name]
    c, ref, got, infl, status, _, _ = run_both(k, nv, 24, 600, 150, seed=77 + k + nv, cfg=cfg)
    assert (status == 0).all(), status
    compare_anal(c, ref, got, k, nv, bool(cfg.get("det_run", 0)))
    assert np.abs(infl - ref["infl"]).max() <= 1e-12
    assert ("letkf_point_kernel" in ctx().last_path()) == (nv + 2 > 16), ctx().last_path()


@pytest.mark.parametrize("k", [17, 18, 19, 49, 51, 62, 16, 21])
def test_das_points_ensemble_sizes_inside_an_instantiation(k):
    """Ensemble sizes that do not fill their instantiation of the wave kernel (KR = 20 serves 17 .. 20, KR = 50 serves 49 and 50,
    KR = 64 serves 51 .. 62): the narrow last block of the Gram -- on the vector ALU at KR = 20 / 50 (r4) -- carries fewer members
    than the instantiation allows for, the departure columns sit elsewhere.  DET_RUN + adaptive inflation, against the oracle; and
    the letkf_core batch entry (mode 1: dense hdxb) for the same sizes through its golden-style check."""
    cfg = CONFIGS["rtps_adaptive_det"]
    c, ref, got, infl, status, _, _ = run_both(k, 11, 36, 500, 120, seed=900 + k, cfg=cfg)
    assert (status == 0).all(), status
    compare_anal(c, ref, got, k, 11, True)
    assert np.abs(infl - ref["infl"]).max() <= 1e-12
    # the same with the k x k outputs requested (the KKOUT instantiation): T and w-bar against the oracle
    c, ref, got, infl, status, trans, transm = run_both(k, 11, 12, 300, 90, seed=950 + k, cfg=CONFIGS["rtps"], want_trans=True)
    assert (status == 0).all(), status
    t = trans.cpu().numpy()
    for p in range(12):
        e = ref["trans"][p]
        assert np.abs(t[p] - e).max() <= 1e-11 * np.abs(e).max(), (p, np.abs(t[p] - e).max())
    assert np.abs(transm.cpu().numpy() - ref["transm"]).max() <= 1e-11 * max(np.abs(ref["transm"]).max(), 1e-300)


@pytest.mark.parametrize("seed", range(8))
def test_das_points_small_ensembles_soak(seed):
    """Randomised soak of the three-points-per-wave kernel (k = 2 .. 20) against the oracle: observation errors from 0.05 to 30
    (cond(A) up to ~1e5), local lists from empty to several staging batches (> 256), batches that are not multiples of three runs,
    random run lengths, every relaxation / inflation switch."""
    from _gpu import ctx, dev
    rng = np.random.default_rng(4000 + seed)
    k = int(rng.integers(2, 21))
    npts = int(rng.integers(40, 230))
    n_mean = int(rng.choice([4, 50, 170, 330]))
    nobs_tot = 720
    name = list(CONFIGS)[int(rng.integers(0, len(CONFIGS)))]
    cfg = CONFIGS[name]
    det = bool(cfg.get("det_run", 0))
    c = das_case(k=k, nv=11, npts=npts, nobs_tot=nobs_tot, n_mean=n_mean, seed=4100 + seed, det_run=det, infl0=1.05)
    f = rng.choice([0.05, 1.0, 6.0], size=nobs_tot)
    c["rdiag"] = c["rdiag"] * f[c["obs_idx"]] ** 2
    prm = _oracle.DasParams(k=k, nv=11, det_run=int(det), infl_adaptive=cfg.get("infl_adaptive", 0),
                            relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0), relax_alpha=cfg.get("relax_alpha", 0.0),
                            relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0), q_update_top=cfg.get("q_update_top", 0.0),
                            q_sprd_max=cfg.get("q_sprd_max", 0.0), iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=4)
    ref = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], c["rdiag"], c["rloc"], c["ensval"], c["dep"], c["beta"], c["infl"],
                             c["gues"], c["sp"], c["sm"], c["sv"])
    assert ref["rc"] == 0
    anal = torch.full((c["gues"].size,), float("nan"), dtype=torch.float64, device="cuda")
    infl = dev(c["infl"])
    status = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    nsweep = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    ctx().das_points(k, 11, dev(c["obs_off"]), dev(c["obs_idx"]), dev(c["rdiag"]), dev(c["rloc"]), dev(c["ensval"]), c["kld"],
                     dev(c["dep"]), infl, dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"], beta=dev(c["beta"]), det_run=det,
                     infl_adaptive=cfg.get("infl_adaptive", 0), relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                     relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                     q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0), iv_p=4, iv_q_first=5,
                     iv_q_last=10, status=status, nsweep=nsweep, warm_run=int(rng.choice([0, 1, 2, 5, 16])))
    torch.cuda.synchronize()
    assert ctx().last_path().startswith("letkf_trio_kernel"), ctx().last_path()
    st = status.cpu().numpy()
    assert set(st.tolist()) <= {0, 3}, st                     # (3: the conditioning warning of common_mtx.f90:66-78 -- the analysis is still made)
    live = (np.diff(c["obs_off"]) > 0) & (c["beta"] != 0.0)
    sw = nsweep.cpu().numpy()
    assert (sw[live] >= 1).all() and (sw[live] < 40).all() and (sw[~live] == 0).all()
    compare_anal(c, ref, anal.cpu().numpy(), k, 11, det)
    assert np.abs(infl.cpu().numpy() - ref["infl"]).max() <= 1e-11


@pytest.mark.parametrize("k,trio", [(20, 1), (20, 0), (17, 1), (50, 1)])
def test_a_point_with_a_nan_observation_poisons_no_other_point(k, trio):
    """One observation row whose ensemble perturbations are NaN, in the list of ONE grid point: that point's analysis is NaN (as the
    reference's would be), every other point's is what it is without the row -- also the points solved beside it in the same wave
    (three per wave at k <= 20: their padded matrix operands read past their own park) and the next point of its run."""
    from _gpu import ctx, dev
    npts = 90
    c = das_case(k=k, nv=11, npts=npts, nobs_tot=400, n_mean=60, seed=8800 + k, det_run=False, infl0=1.02)
    n = np.diff(c["obs_off"])
    victim = int(np.argmax((n > 20) & (c["beta"] == 1.0) & (np.arange(npts) > 30)))
    ens_bad = np.vstack([c["ensval"].reshape(-1, c["kld"]), np.full((1, c["kld"]), np.nan)])
    dep_bad = np.concatenate([c["dep"], [0.5]])
    idx_bad = c["obs_idx"].copy()
    idx_bad[c["obs_off"][victim]] = ens_bad.shape[0] - 1

    def run(ens, dep, idx):
        anal = torch.full((c["gues"].size,), float("nan"), dtype=torch.float64, device="cuda")
        infl = dev(c["infl"])
        ctx().das_points(k, 11, dev(c["obs_off"]), dev(idx), dev(c["rdiag"]), dev(c["rloc"]), dev(np.ascontiguousarray(ens).reshape(-1)),
                         c["kld"], dev(dep), infl, dev(c["gues"]), anal, c["sp"], c["sm"], c["sv"], beta=dev(c["beta"]),
                         relax_alpha_spread=0.95, warm_run=8)
        torch.cuda.synchronize()
        return anal.cpu().numpy().reshape(11, c["nens"], npts)[:, :k]
    ctx().set_option(ctx().OPT_SMALL_K_TRIO, trio)
    try:
        clean = run(c["ensval"].reshape(-1, c["kld"]), c["dep"], c["obs_idx"])
        bad = run(ens_bad, dep_bad, idx_bad)
    finally:
        ctx().set_option(ctx().OPT_SMALL_K_TRIO, 1)
    others = np.arange(npts) != victim
    assert np.isnan(bad[:, :, victim]).all()
    assert np.isfinite(bad[:, :, others]).all(), np.argwhere(~np.isfinite(bad[:, :, others]).all(axis=(0, 1))).ravel()
    scale = np.abs(clean[:, :, others]).max(axis=(1, 2), keepdims=True)
    assert (np.abs(bad[:, :, others] - clean[:, :, others]) / scale).max() <= 1e-11
