"""obs_local fused into the loop body (letkf_das_points_fused_dev) against the two-step path it replaces
(letkf_obs_search_dev lists -> letkf_das_points_dev): same candidate order, same 4-obs MFMA grouping, so the analysis
and status must be BIT-identical, the adaptive inflation to the last place or one unit of it (points without observations: equal to rounding, they take different
kernels), and nobs_out must be the list lengths; the two-step path itself is checked
against the oracle elsewhere (test_gpu_search.py, test_gpu_das.py)."""
import numpy as np
import pytest
import torch

from _search import build_case, device_struct

pytestmark = pytest.mark.gpu


def _state(k, nv, npts, det, seed):
    rng = np.random.default_rng(seed)
    nens = k + 1 + (1 if det else 0)
    gues = rng.normal(0.0, 1.0, (nv, nens, npts))
    gues[:, :k] -= gues[:, :k].mean(axis=1, keepdims=True)       # perturbations in the member slots
    gues[:, k] = rng.normal(5.0, 1.0, (nv, npts))                 # mean slot
    gues[4, k] = rng.uniform(3.0e4, 1.0e5, npts)                  # pressure
    return np.ascontiguousarray(gues.reshape(-1)), nens


@pytest.mark.parametrize("k,det,cfg", [(50, False, dict(relax_alpha_spread=0.95)),
                                       (20, True, dict(relax_alpha_spread=0.9, infl_adaptive=1, relax_to_inflated_prior=1)),
                                       (33, True, dict(relax_alpha=0.6, q_update_top=6.0e4)),
                                       (62, False, dict()),
                                       (16, True, dict(relax_alpha_spread=0.95, infl_adaptive=1)),
                                       (17, False, dict(relax_alpha_spread=0.95)),
                                       (30, True, dict(relax_alpha_spread=0.9)),
                                       (51, False, dict(relax_alpha=0.5))])
def test_fused_search_equals_search_then_solve(k, det, cfg):
    from _gpu import ctx, dev
    c = ctx()
    case = build_case(21 + k, npts=230)
    t, keep = device_struct(case, "cuda")
    nobs, nv, npts = case["nobs"], 11, 230
    kld = k + (1 if det else 0)
    rng = np.random.default_rng(k)
    ensval = rng.normal(0.0, 2.0, (nobs, kld))
    ensval[:, :k] -= ensval[:, :k].mean(axis=1, keepdims=True)
    dep = rng.normal(0.0, 3.0, nobs)
    p = case["pts"]
    pts = [dev(p[f]) for f in ("ri", "rj", "rlev", "rz")]
    gues, nens = _state(k, nv, npts, det, k)
    beta = np.ones(npts)
    beta[::17] = 0.0
    beta[5::23] = 0.4
    sp, sm, sv = 1, npts, npts * nens
    off, idx, rd, rl = c.obs_search(t, *pts)
    out = {}
    for mode in ("lists", "fused"):
        anal = torch.full((gues.size,), float("nan"), dtype=torch.float64, device="cuda")
        infl = torch.full((npts * nv,), 1.03, dtype=torch.float64, device="cuda")
        st = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
        nob = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
        kw = dict(beta=dev(beta), det_run=det, status=st, iv_p=4, iv_q_first=5, iv_q_last=10, warm_run=5, **cfg)
        if mode == "lists":
            c.set_option(c.OPT_SMALL_K_TRIO, 0)  # (k <= 20: the list route of the SAME kernel, not three points per wave)
            try:
                c.das_points(k, nv, off, idx, rd, rl, dev(ensval), kld, dev(dep), infl, dev(gues), anal, sp, sm, sv, **kw)
            finally:
                c.set_option(c.OPT_SMALL_K_TRIO, 1)
        else:
            c.das_points(k, nv, None, None, None, None, dev(ensval), kld, dev(dep), infl, dev(gues), anal, sp, sm, sv,
                         fused=(t, *pts), nobs_out=nob, **kw)
        torch.cuda.synchronize()
        out[mode] = (anal, infl, st, nob)
    a0, i0, s0, _ = out["lists"]
    a1, i1, s1, n1 = out["fused"]
    assert int(s0.abs().max()) == 0 and int(s1.abs().max()) == 0
    counts = (off[1:] - off[:-1]).to(torch.int32)
    live = torch.from_numpy(beta != 0.0).cuda()
    assert torch.equal(n1[live], counts[live]) and int(n1[~live].abs().max()) == 0
    assert int(counts.max()) > 256 and int(counts.min()) < 192      # points with one and with several staging batches
    m0, m1 = a0.view(nv, nens, npts), a1.view(nv, nens, npts)
    members = list(range(k)) + ([k + 1] if det else [])
    # points with observations (and beta = 0 points: a copy): bit for bit.  Points WITHOUT observations leave the list
    # path through the streaming pass (letkf_trivial.hip) and the fused path through the solve kernel's closed-form
    # branch: the same formulas in two kernels, equal to rounding
    empty = live & (counts == 0)
    assert torch.equal(m0[:, members][:, :, ~empty], m1[:, members][:, :, ~empty])
    if int(empty.sum()) > 0:
        e0, e1 = m0[:, members][:, :, empty], m1[:, members][:, :, empty]
        assert float(((e0 - e1).abs() / e1.abs().clamp_min(1.0)).max()) <= 4e-16 * 8
    # the adaptive inflation takes sum(rloc) (parm(3), common_letkf.f90:233) from per-lane partial sums: the fused search packs a
    # point's second and later staging batches behind the left-over entries of the one before, the list path does not -- the
    # same numbers summed in another order, one unit in the last place on a point in a few hundred
    assert float((i0 - i1).abs().max()) <= 4.5e-16, (float((i0 - i1).abs().max()), int((i0 != i1).sum()), i0.numel())


def test_fused_search_refuses_what_it_does_not_cover():
    from _gpu import ctx, dev
    c = ctx()
    case = build_case(3, max_nobs=(25, 25, 10, 5), npts=8)
    t, keep = device_struct(case, "cuda")
    p = case["pts"]
    pts = [dev(p[f]) for f in ("ri", "rj", "rlev", "rz")]
    k, nv, npts = 20, 11, 8
    gues, nens = _state(k, nv, npts, False, 1)
    z = torch.zeros(case["nobs"], k, dtype=torch.float64, device="cuda")
    args = (dev(np.zeros(case["nobs"])), torch.ones(npts * nv, dtype=torch.float64, device="cuda"), dev(gues),
            torch.zeros(gues.size, dtype=torch.float64, device="cuda"), 1, npts, npts * nens)
    with pytest.raises(RuntimeError):                     # MAX_NOBS_PER_GRID > 0
        c.das_points(k, nv, None, None, None, None, z, k, *args, fused=(t, *pts))
    case2 = build_case(3, npts=8)
    t2, keep2 = device_struct(case2, "cuda")
    gues100, nens100 = _state(100, nv, npts, False, 1)
    z100 = torch.zeros(case2["nobs"], 100, dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError):                     # two-wave kernel: lists only
        c.das_points(100, nv, None, None, None, None, z100, 100, dev(np.zeros(case2["nobs"])),
                     torch.ones(npts * nv, dtype=torch.float64, device="cuda"), dev(gues100),
                     torch.zeros(gues100.size, dtype=torch.float64, device="cuda"), 1, npts, npts * nens100,
                     fused=(t2, *pts))
