"""letkf_das_columns_dev (C ABI 3c): das_letkf's main loop for a whole subdomain in one call -- the column search and the loop
body by slabs of levels whose local-observation lists fit a workspace of the library (the reference's level loop,
scale/letkf/letkf_tools.f90:313) -- against the two-call route (letkf_obs_search_columns_dev for all levels, then ONE
letkf_das_points_dev): same local observations at every point, same analysis to rounding (the slabs only change which
points warm-start from which), the adaptively updated inflation field and the RTPS diagnostic at the right places of the
WHOLE field (letkf_das_args.infl_sv), for slab sizes from one level to all of them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,list_mb", [("C2-mini", 0.5), ("C2-mini", 3.0), ("C2-mini", 0), ("C2-mini-k100", 2.0), ("C2-mini-disc", 1.0),
                                          ("C2-mini-k20", 0.4), ("C2-tiny-k17", 0.1), ("C2-tiny-k16", 0)])
def test_column_pipeline_equals_search_plus_loop_body(name, list_mb):
    import bench_workload as bw
    from _gpu import ctx, pkg
    dev = torch.device("cuda:0")
    w = bw.build(name, dev, det_run=True)
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    nij1, nlev = w["cfg"]["nx"] * w["cfg"]["ny"], w["cfg"]["nz"]
    cx = ctx()
    cx.ens_mean(k, nv, npts, w["gues"], 1, npts, npts * nens)
    cx.to_perturbations(k, nv, npts, w["gues"], 1, npts, npts * nens)
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    ens, dep = w["ensval"][order].contiguous(), w["dep"][order].contiguous()
    rig, rjg = pts[0][:nij1].contiguous(), pts[1][:nij1].contiguous()
    sw = dict(det_run=True, infl_adaptive=True, relax_alpha_spread=0.95)

    def fresh():
        return (torch.full_like(w["gues"], float("nan")), torch.full((npts * nv,), 1.03, dtype=torch.float64, device=dev),
                torch.full((npts,), -1, dtype=torch.int32, device=dev), torch.zeros(npts * nv, dtype=torch.float64, device=dev))
    # two calls: all lists, then the loop body
    a0, i0, s0, r0 = fresh()
    off, idx, rd, rl = cx.obs_search_columns(t_s, nij1, nlev, rig, rjg, pts[2], pts[3])
    cx.das_points(k, nv, off, idx, rd, rl, ens, w["kld"], dep, i0, w["gues"], a0, 1, npts, npts * nens, status=s0,
                  rtps_infl_out=r0, **sw)
    # one call: the list-free route (survivors per column, batches of columns) where the one-wave kernel serves the call, and
    # the lists by slabs of levels inside the library (LETKF_OPT_COLUMN_SURVIVORS = 0; k = 100 takes them anyway)
    for survivors in (1, 0):
        a1, i1, s1, r1 = fresh()
        nobs = torch.full((npts,), -7, dtype=torch.int32, device=dev)
        cx.set_option(cx.OPT_COLUMN_SURVIVORS, survivors)
        try:
            cx.das_columns(k, nv, t_s, nij1, nlev, rig, rjg, pts[2], pts[3], ens, w["kld"], dep, i1, w["gues"], a1, 1, npts,
                           npts * nens, list_bytes=int(list_mb * 2 ** 20), nobs_out=nobs, status=s1, rtps_infl_out=r1, **sw)
            torch.cuda.synchronize()
        finally:
            cx.set_option(cx.OPT_COLUMN_SURVIVORS, 2)
        assert ("FUSED" in cx.last_path()) == (survivors == 1 and k <= 62), cx.last_path()
        assert int(s0.abs().max()) == 0 and int(s1.abs().max()) == 0
        assert torch.equal(nobs.long(), off[1:] - off[:-1])
        g0 = a0.view(nv, nens, npts)
        g1 = a1.view(nv, nens, npts)
        x = w["gues"].view(nv, nens, npts)
        members = list(range(k)) + [k + 1]
        for v in range(nv):
            scale = float(max(x[v, k].abs().max(), x[v, :k].abs().max()))
            assert float((g0[v, members] - g1[v, members]).abs().max()) <= 1e-11 * scale, v
        assert float((i0 - i1).abs().max()) <= 1e-12
        assert float((r0 - r1).abs().max()) <= 1e-11
        assert bool(torch.isnan(g1[:, k]).all())              # the mean slot is not the loop body's to write


@pytest.mark.parametrize("name", ["C2-mini", "C2-mini-disc", "C2-mini-k20", "C2-tiny-k16", "C2-tiny-k17", "C2-tiny-k30", "C2-tiny-k40", "C2-tiny-k51", "C2-tiny-k60"])
def test_list_free_route_equals_the_lists_bit_for_bit(name):
    """Same runs (up whole columns), same weights (search_dev::column_vertical_cal is the column search's arithmetic), same order
    of the local observations: the analysis of the list-free route IS the list route's, to the last bit."""
    import bench_workload as bw
    from _gpu import ctx, pkg
    dev = torch.device("cuda:0")
    w = bw.build(name, dev)
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    nij1, nlev = w["cfg"]["nx"] * w["cfg"]["ny"], w["cfg"]["nz"]
    cx = ctx()
    cx.ens_mean(k, nv, npts, w["gues"], 1, npts, npts * nens)
    cx.to_perturbations(k, nv, npts, w["gues"], 1, npts, npts * nens)
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    ens, dep = w["ensval"][order].contiguous(), w["dep"][order].contiguous()
    rig, rjg = pts[0][:nij1].contiguous(), pts[1][:nij1].contiguous()
    off, idx, rd, rl = cx.obs_search_columns(t_s, nij1, nlev, rig, rjg, pts[2], pts[3])
    a0 = torch.full_like(w["gues"], float("nan"))
    i0 = torch.ones(npts * nv, dtype=torch.float64, device=dev)
    cx.set_option(cx.OPT_SMALL_K_TRIO, 0)        # (k <= 20: the list route of the SAME kernel, not three points per wave)
    try:
        cx.das_points(k, nv, off, idx, rd, rl, ens, w["kld"], dep, i0, w["gues"], a0, 1, npts, npts * nens, relax_alpha_spread=0.95,
                      warm_stride=nij1)
    finally:
        cx.set_option(cx.OPT_SMALL_K_TRIO, 1)
    a1 = torch.full_like(w["gues"], float("nan"))
    i1 = torch.ones(npts * nv, dtype=torch.float64, device=dev)
    cx.set_option(cx.OPT_COLUMN_SURVIVORS, 1)
    try:
        cx.das_columns(k, nv, t_s, nij1, nlev, rig, rjg, pts[2], pts[3], ens, w["kld"], dep, i1, w["gues"], a1, 1, npts,
                       npts * nens, list_bytes=1 << 34, relax_alpha_spread=0.95)
        torch.cuda.synchronize()
    finally:
        cx.set_option(cx.OPT_COLUMN_SURVIVORS, 2)
    assert "FUSED" in cx.last_path()
    g0, g1 = a0.view(nv, nens, npts)[:, :k], a1.view(nv, nens, npts)[:, :k]
    assert torch.equal(g0, g1)


def test_the_route_is_chosen_by_the_size_of_the_lists():
    """LETKF_OPT_COLUMN_SURVIVORS = 2 (default): lists where they fit list_bytes at once, the list-free route where they do not."""
    import bench_workload as bw
    from _gpu import ctx, pkg
    dev = torch.device("cuda:0")
    w = bw.build("C2-mini", dev)
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    nij1, nlev = w["cfg"]["nx"] * w["cfg"]["ny"], w["cfg"]["nz"]
    cx = ctx()
    cx.ens_mean(k, nv, npts, w["gues"], 1, npts, npts * nens)
    cx.to_perturbations(k, nv, npts, w["gues"], 1, npts, npts * nens)
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    ens, dep = w["ensval"][order].contiguous(), w["dep"][order].contiguous()
    rig, rjg = pts[0][:nij1].contiguous(), pts[1][:nij1].contiguous()
    res = []
    for lb, fused in ((1 << 34, False), (1 << 20, True)):
        a1 = torch.full_like(w["gues"], float("nan"))
        i1 = torch.ones(npts * nv, dtype=torch.float64, device=dev)
        cx.das_columns(k, nv, t_s, nij1, nlev, rig, rjg, pts[2], pts[3], ens, w["kld"], dep, i1, w["gues"], a1, 1, npts, npts * nens,
                       list_bytes=lb, relax_alpha_spread=0.95)
        torch.cuda.synchronize()
        assert ("FUSED" in cx.last_path()) == fused, (lb, cx.last_path())
        res.append(a1.view(nv, nens, npts)[:, :k].clone())
    x = w["gues"].view(nv, nens, npts)
    for v in range(nv):
        scale = float(max(x[v, k].abs().max(), x[v, :k].abs().max()))
        assert float((res[0][v] - res[1][v]).abs().max()) <= 1e-11 * scale


@pytest.mark.parametrize("nlev,list_kb", [(5, 1 << 20), (9, 64)])
def test_list_free_route_with_four_combined_types_and_three_vertical_modes(nlev, list_kb):
    """The survivors of a column carry their combined type: a merged radar group (height), upper-air T (ln p) and surface pressure
    (the observed value as vertical coordinate), different horizontal and vertical scales and variable-localisation factors
    (tests/_search.py) -- the type's numbers are read per 64-entry chunk, every type's entries padded to whole chunks.  Against
    letkf_obs_search_columns_dev + ONE letkf_das_points_dev with the same runs up the columns: bit for bit; small workspaces force
    several batches of columns."""
    from _gpu import ctx, dev
    from _search import build_case, device_struct
    nij1 = 60
    case = build_case(91, npts=nij1)
    t, keep = device_struct(case, "cuda")
    p = case["pts"]
    rng = np.random.default_rng(7 + nlev)
    k, nv, nens = 24, 11, 25
    npts = nij1 * nlev
    nobs = len(case["arr"]["ob_ri"]) if "ob_ri" in case["arr"] else int(case["ctype_rows"][-1])
    rlev = rng.uniform(2.5e4, 1.0e5, npts)
    rz = rng.uniform(0.0, 12000.0, npts)
    kld = k + 1
    ens = rng.standard_normal((nobs, kld)) * 2.0
    ens[:, :k] -= ens[:, :k].mean(axis=1, keepdims=True)
    dep = rng.standard_normal(nobs) * 3.0
    gues = rng.standard_normal(nv * nens * npts)
    gv = gues.reshape(nv, nens, npts)
    gv[:, :k] -= gv[:, :k].mean(axis=1, keepdims=True)          # perturbations in slots 0..k-1, the mean in slot k
    gv[:, k] = 10.0 + rng.standard_normal((nv, npts))
    c = ctx()
    d = lambda a: dev(np.ascontiguousarray(a))
    g_ens, g_dep, g_gues = d(ens.reshape(-1)), d(dep), d(gues)
    rig, rjg, g_rlev, g_rz = d(p["ri"][:nij1]), d(p["rj"][:nij1]), d(rlev), d(rz)
    off, idx, rd, rl = c.obs_search_columns(t, nij1, nlev, rig, rjg, g_rlev, g_rz)
    assert int(off[-1]) > 20 * npts
    a0 = torch.full((gues.size,), float("nan"), dtype=torch.float64, device="cuda")
    i0 = torch.ones(npts * nv, dtype=torch.float64, device="cuda")
    s0 = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    c.das_points(k, nv, off, idx, rd, rl, g_ens, kld, g_dep, i0, g_gues, a0, 1, npts, npts * nens, relax_alpha_spread=0.9,
                 status=s0, warm_stride=nij1)
    a1 = torch.full((gues.size,), float("nan"), dtype=torch.float64, device="cuda")
    i1 = torch.ones(npts * nv, dtype=torch.float64, device="cuda")
    s1 = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
    nobs_out = torch.full((npts,), -5, dtype=torch.int32, device="cuda")
    c.set_option(c.OPT_COLUMN_SURVIVORS, 1)
    try:
        c.das_columns(k, nv, t, nij1, nlev, rig, rjg, g_rlev, g_rz, g_ens, kld, g_dep, i1, g_gues, a1, 1, npts, npts * nens,
                      list_bytes=list_kb * 1024, nobs_out=nobs_out, relax_alpha_spread=0.9, status=s1)
        torch.cuda.synchronize()
    finally:
        c.set_option(c.OPT_COLUMN_SURVIVORS, 2)
    assert "FUSED" in c.last_path()
    assert int(s0.abs().max()) == 0 and int(s1.abs().max()) == 0
    assert torch.equal(nobs_out.long(), off[1:] - off[:-1])
    g0, g1 = a0.view(nv, nens, npts)[:, :k], a1.view(nv, nens, npts)[:, :k]
    if list_kb >= 1 << 20:
        assert torch.equal(g0, g1)                   # one batch: the same runs, the same bits
    else:
        # (batches of a few columns: the launch splits its runs differently -- other warm starts, same analysis to rounding)
        assert float((g0 - g1).abs().max()) <= 1e-11 * float(g0.abs().max())


def _oracle_sample_check(w, t_s, keep, pts4, ens, dep, sample, beta, infl0, anal, infl1, nobs, sw, tol=1e-10):
    """The analysis of the grid points `sample` against the ORACLE end to end: obs_local by oracle/letkf_oracle.c orc_obs_local
    on a host copy of the same tables (scale/letkf/letkf_tools.f90:1325-1759 -- nothing of the device search is involved), then
    the loop body orc_das_letkf_points (:313-527) on those lists.  Returns the worst |d xa| / max(|x-bar|, |x'|)."""
    import _oracle
    import _search
    import bench_workload as bw
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    h, alive = _search.host_struct_from_torch(t_s, keep)
    P = [p.cpu().numpy() for p in pts4]
    off, idx, rd, rl, tied = _search.oracle_csr(h, P[0][sample], P[1][sample], P[2][sample], P[3][sample])
    assert not tied.any()
    ns = len(sample)
    tp = torch.from_numpy(sample).cuda()
    gs = bw.state_view(w, w["gues"])[:, :, tp].contiguous().cpu().numpy().reshape(-1)
    b = beta[tp].cpu().numpy() if beta is not None else None
    i0 = infl0.view(nv, npts)[:, tp].contiguous().cpu().numpy().reshape(-1)
    prm = _oracle.DasParams(k=k, nv=nv, det_run=int(sw.get("det_run", False)), infl_adaptive=int(sw.get("infl_adaptive", False)),
                            relax_to_inflated_prior=0, relax_alpha=sw.get("relax_alpha", 0.0),
                            relax_alpha_spread=sw.get("relax_alpha_spread", 0.0), q_update_top=0.0, q_sprd_max=0.0, iv_p=4,
                            iv_q_first=5, iv_q_last=10, nthreads=8)
    ref = _oracle.das_points(prm, off, idx, rd, rl, ens.cpu().numpy(), dep.cpu().numpy(), b, i0, gs, 1, ns, ns * nens)
    assert ref["rc"] == 0
    got = bw.state_view(w, anal)[:, :, tp].cpu().numpy()
    exp = ref["anal"].reshape(nv, nens, ns)
    x = gs.reshape(nv, nens, ns)
    members = list(range(k)) + ([k + 1] if sw.get("det_run") else [])
    worst = 0.0
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        assert np.isfinite(got[v, members]).all()
        err = np.abs(got[v, members] - exp[v, members]).max()
        assert err <= tol * scale, (v, err, scale)
        worst = max(worst, err / scale)
    gi = infl1.view(nv, npts)[:, tp].cpu().numpy().reshape(-1)
    assert np.abs(gi - ref["infl"]).max() <= 1e-12
    if nobs is not None:
        cnt = np.diff(off)
        live = np.ones(ns, bool) if b is None else b != 0.0
        assert np.array_equal(nobs[tp].cpu().numpy()[live], cnt[live])
    return worst, np.diff(off)


@pytest.mark.parametrize("name,list_mb,nsample", [("C4-slab", 24.0, 20), ("C4-slab-disc", 40.0, 28)])
def test_list_free_route_at_configs3_density_against_the_oracle(name, list_mb, nsample):
    """BASELINE configs[3]'s density (n ~ 4900 local observations, ~16 k horizontal survivors per column: dozens of 256-entry
    passes, both halves of a wave's list slot, slots of sl_cap entries) on the route `das_letkf_amd` takes there
    (LETKF_OPT_COLUMN_SURVIVORS = 1, two levels per pass), with DET_RUN, adaptive inflation and a beta field that has zeros
    and fractions, in batches of columns whose last one is ragged: sampled points against the oracle's obs_local + loop body."""
    import bench_workload as bw
    from _gpu import ctx, pkg
    dev = torch.device("cuda:0")
    w = bw.build(name, dev, det_run=True, lists=False)
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    nij1, nlev = w["cfg"]["nx"] * w["cfg"]["ny"], w["cfg"]["nz"]
    cx = ctx()
    cx.ens_mean(k, nv, npts, w["gues"], 1, npts, npts * nens)
    cx.to_perturbations(k, nv, npts, w["gues"], 1, npts, npts * nens)
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    ens, dep = w["ensval"][order].contiguous(), w["dep"][order].contiguous()
    rig, rjg = pts[0][:nij1].contiguous(), pts[1][:nij1].contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    beta = torch.rand(npts, generator=g, device=dev, dtype=torch.float64)
    beta[torch.rand(npts, generator=g, device=dev) < 0.15] = 0.0       # relax_beta's zeros (letkf_tools.f90:333-359)
    beta[torch.rand(npts, generator=g, device=dev) < 0.5] = 1.0
    infl0 = 1.0 + 0.1 * torch.rand(npts * nv, generator=g, device=dev, dtype=torch.float64)
    sw = dict(det_run=True, infl_adaptive=True, relax_alpha_spread=0.95)
    anal = torch.full_like(w["gues"], float("nan"))
    infl = infl0.clone()
    status = torch.full((npts,), -1, dtype=torch.int32, device=dev)
    nobs = torch.full((npts,), -7, dtype=torch.int32, device=dev)
    cx.set_option(cx.OPT_COLUMN_SURVIVORS, 1)
    try:
        cx.das_columns(k, nv, t_s, nij1, nlev, rig, rjg, pts[2], pts[3], ens, w["kld"], dep, infl, w["gues"], anal, 1, npts,
                       npts * nens, list_bytes=int(list_mb * 2 ** 20), nobs_out=nobs, beta=beta, status=status, **sw)
        torch.cuda.synchronize()
    finally:
        cx.set_option(cx.OPT_COLUMN_SURVIVORS, 2)
    assert "FUSED" in cx.last_path()
    assert int(status.abs().max()) == 0
    # the sample: points of the densest columns, of the disc's rim, of empty columns (disc), beta = 0 / fraction / 1
    rng = np.random.default_rng(5)
    nb = nobs.cpu().numpy()
    dense = np.flatnonzero(nb >= 0.9 * nb.max())
    bz = beta.cpu().numpy()
    sample = [rng.choice(dense, size=min(len(dense), nsample // 2), replace=False), rng.choice(npts, size=nsample // 2, replace=False),
              rng.choice(np.flatnonzero(bz == 0.0), size=2, replace=False), rng.choice(np.flatnonzero(bz == 1.0), size=2, replace=False)]
    if "disc" in name:
        rim = np.flatnonzero((nb > 0) & (nb < 600))
        empty = np.flatnonzero((nb == 0) & (beta.cpu().numpy() != 0.0))
        assert len(rim) and len(empty)
        sample += [rng.choice(rim, size=4, replace=False), rng.choice(empty, size=2, replace=False)]
    sample = np.unique(np.concatenate(sample)).astype(np.int64)
    worst, cnt = _oracle_sample_check(w, t_s, keep, pts, ens, dep, sample, beta, infl0, anal, infl, nobs, sw)
    assert cnt.max() > 4500 and (cnt % 256 != 0).any()
    bs = beta.cpu().numpy()[sample]
    assert (bs == 0.0).any() and (bs == 1.0).any() and ((bs > 0.0) & (bs < 1.0)).any()
    # the geometry the test is about: several batches of columns with a ragged last one, columns of many passes
    print(f"{name}: {len(sample)} points against the oracle, worst {worst:.2e}, n up to {cnt.max()}")


@pytest.mark.parametrize("nlev_cut,run", [(11, 0), (7, 5), (12, 3)])
def test_list_free_route_with_beta_zeros_and_odd_runs_equals_the_lists(nlev_cut, run):
    """The pairs of levels of the two-levels-per-pass pre-pass under everything that breaks a pair: an odd number of levels,
    runs of an odd length (warm_run) and quarter-split units, beta = 0 at the first / second point of a pair (the pass is
    skipped, or its second list is dropped).  Against the list route with the same runs: same lists, same analysis."""
    import bench_workload as bw
    from _gpu import ctx, pkg
    dev = torch.device("cuda:0")
    w = bw.build("C2-mini", dev)
    k, nv, nens = w["k"], w["nv"], w["nens"]
    nij1, nlev_all = w["cfg"]["nx"] * w["cfg"]["ny"], w["cfg"]["nz"]
    npts_all = w["npts"]
    nlev = nlev_cut
    npts = nij1 * nlev
    cx = ctx()
    cx.ens_mean(k, nv, npts_all, w["gues"], 1, npts_all, npts_all * nens)
    cx.to_perturbations(k, nv, npts_all, w["gues"], 1, npts_all, npts_all * nens)
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    ens, dep = w["ensval"][order].contiguous(), w["dep"][order].contiguous()
    rig, rjg = pts[0][:nij1].contiguous(), pts[1][:nij1].contiguous()
    rlev, rz = pts[2][:npts].contiguous(), pts[3][:npts].contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(3 + nlev)
    beta = torch.rand(npts, generator=g, device=dev, dtype=torch.float64)
    beta[torch.rand(npts, generator=g, device=dev) < 0.3] = 0.0
    beta[torch.rand(npts, generator=g, device=dev) < 0.3] = 1.0
    beta.view(nlev, nij1)[:, :7] = 0.0                                  # whole columns switched off
    beta.view(nlev, nij1)[0::2, 7:14] = 0.0                             # every first / every second point of a pair
    beta.view(nlev, nij1)[1::2, 14:21] = 0.0
    off, idx, rd, rl = cx.obs_search_columns(t_s, nij1, nlev, rig, rjg, rlev, rz)
    # (the state keeps the strides of the full workload: element (p, m, v) at p + m npts_all + v npts_all nens)
    a0 = torch.full_like(w["gues"], float("nan"))
    i0 = torch.ones(npts_all * nv, dtype=torch.float64, device=dev)
    s0 = torch.full((npts,), -1, dtype=torch.int32, device=dev)
    cx.das_points(k, nv, off, idx, rd, rl, ens, w["kld"], dep, i0, w["gues"], a0, 1, npts_all, npts_all * nens, beta=beta,
                  relax_alpha_spread=0.95, infl_adaptive=True, status=s0, warm_stride=nij1, warm_run=run, infl_sv=npts_all)
    a1 = torch.full_like(w["gues"], float("nan"))
    i1 = torch.ones(npts_all * nv, dtype=torch.float64, device=dev)
    s1 = torch.full((npts,), -1, dtype=torch.int32, device=dev)
    nobs = torch.full((npts,), -5, dtype=torch.int32, device=dev)
    cx.set_option(cx.OPT_COLUMN_SURVIVORS, 1)
    try:
        cx.das_columns(k, nv, t_s, nij1, nlev, rig, rjg, rlev, rz, ens, w["kld"], dep, i1, w["gues"], a1, 1, npts_all,
                       npts_all * nens, list_bytes=1 << 34, nobs_out=nobs, beta=beta, relax_alpha_spread=0.95, infl_adaptive=True,
                       status=s1, warm_run=run, infl_sv=npts_all)
        torch.cuda.synchronize()
    finally:
        cx.set_option(cx.OPT_COLUMN_SURVIVORS, 2)
    assert "FUSED" in cx.last_path()
    assert int(s0.abs().max()) == 0 and int(s1.abs().max()) == 0
    live = beta != 0.0
    cnt = (off[1:] - off[:-1]).to(torch.int32)
    assert torch.equal(nobs[live], cnt[live]) and int(nobs[~live].abs().max()) == 0
    g0 = a0.view(nv, nens, npts_all)[:, :k, :npts]
    g1 = a1.view(nv, nens, npts_all)[:, :k, :npts]
    assert torch.equal(g0, g1)
    assert torch.equal(i0, i1)
