"""Deterministic synthetic inputs for the letkf_core boundary and the batched das_letkf point update.
Shared by the parity tests, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py."""
import hashlib

import numpy as np

DIST_ZERO_FAC_SQUARE = float(np.float32(13.33333333))  # scale/letkf/letkf_obs.f90:28


def core_case(k, n, seed, nobs=None, cond="benign", rdiag_wloc=True, infl=1.0, with_det=False):
    """One letkf_core problem.  hdxb is (nobs, k) Fortran-ordered with only the first n rows meaningful
    (common/common_letkf.f90:35-37); rows n.. hold a poison value to catch over-reads."""
    rng = np.random.default_rng(seed)
    nobs = max(n, 1) if nobs is None else nobs
    y = rng.standard_normal((n, k))
    y -= y.mean(axis=1, keepdims=True)
    err = np.full(n, 1.0)
    if cond == "ill":  # obs error 1e-3 on a quarter of the obs: cond(A) ~ 1e6
        err[: max(1, n // 4)] = 1e-3
    elif cond == "dup" and n >= 2:  # duplicated obs rows
        y[1::2] = y[0::2][: len(y[1::2])]
    elif cond == "zerocol":
        y[:, 0] = 0.0
    d2 = rng.uniform(0.0, DIST_ZERO_FAC_SQUARE, size=n)
    rloc = np.exp(-0.5 * d2)
    rdiag = err * err / rloc if rdiag_wloc else err * err
    dep = rng.standard_normal(n) * np.sqrt(err * err + 1.0)
    depd = rng.standard_normal(n) * np.sqrt(err * err + 1.0) if with_det else None
    hdxb = np.full((nobs, k), 1.0e30, order="F")
    hdxb[:n, :] = y
    pad = lambda v: np.concatenate([v, np.full(nobs - n, 1.0e30)])
    out = dict(k=k, n=n, nobs=nobs, hdxb=hdxb, rdiag=pad(rdiag), rloc=pad(rloc), dep=pad(dep),
               depd=None if depd is None else pad(depd), infl=float(infl), rdiag_wloc=rdiag_wloc)
    return out


def case_sha(case):
    h = hashlib.sha256()
    n = case["n"]
    for key in ("hdxb", "rdiag", "rloc", "dep", "depd"):
        v = case[key]
        if v is None:
            continue
        v = np.ascontiguousarray(v[:n] if v.ndim == 1 else v[:n, :])
        h.update(v.tobytes())
    return h.hexdigest()


def relerr(a, b):
    """max-norm relative error, SURVEY section 8(c)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


def das_case(k, nv, npts, nobs_tot, n_mean, seed, det_run=False, layout="ref", vary_n=True, infl0=1.0,
             q_mean_pos=True):
    """A batch of grid points for the coarse boundary: an obs table (ensval member-fastest, obsda_sort layout,
    scale/common/common_obs_scale.f90:112-130) + per-point CSR local-obs lists as obs_local would produce them
    + a gues array in the reference's layout gues3d(nij1*nlev, nens, nv3d) flattened point-fastest
    (scale/letkf/letkf_tools.f90:55) holding perturbations, mean (slot k) and det member (slot k+1)."""
    rng = np.random.default_rng(seed)
    nens = k + 1 + (1 if det_run else 0)
    kld = k + 1  # slot k == mmdetobs departure (scale/letkf/letkf_tools.f90:1468)
    ens = rng.standard_normal((nobs_tot, kld))
    ens[:, :k] -= ens[:, :k].mean(axis=1, keepdims=True)
    dep = rng.standard_normal(nobs_tot) * 1.5
    counts = np.full(npts, n_mean, dtype=np.int64)
    if vary_n:
        counts = rng.integers(0, max(2 * n_mean, 1) + 1, size=npts).astype(np.int64)
        counts[rng.integers(0, npts, size=max(1, npts // 8))] = 0  # points with no obs (nobsl == 0 branch)
    counts = np.minimum(counts, nobs_tot)
    off = np.zeros(npts + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    nnz = int(off[-1])
    idx = np.empty(nnz, dtype=np.int32)
    for p in range(npts):
        c = int(counts[p])
        if c:
            idx[off[p]:off[p + 1]] = rng.choice(nobs_tot, size=c, replace=False)
    d2 = rng.uniform(0.0, DIST_ZERO_FAC_SQUARE, size=nnz)
    rloc = np.exp(-0.5 * d2)
    err = rng.choice([1.0, 3.0, 5.0], size=nobs_tot)
    rdiag = err[idx] ** 2 / rloc
    # state: (pt, m, v) at pt + npts*(m + nens*v)
    x = rng.standard_normal((nv, nens, npts))
    x[:, :k, :] *= np.array([2.0, 2.0, 2.0, 1.0, 50.0] + [1e-3] * max(nv - 5, 0))[:nv, None, None]
    mean = rng.standard_normal((nv, npts)) * 5.0 + np.array([5.0] * nv)[:, None] * 10.0
    if q_mean_pos and nv > 5:
        mean[5:] = np.abs(mean[5:]) * 1e-3 + 1e-3
    x[:, :k, :] -= x[:, :k, :].mean(axis=1, keepdims=True)
    x[:, k, :] = mean
    if det_run:
        x[:, k + 1, :] = mean + rng.standard_normal((nv, npts))
    gues = np.ascontiguousarray(x).reshape(-1)
    beta = np.ones(npts)
    beta[rng.integers(0, npts, size=max(1, npts // 10))] = 0.0
    beta[rng.integers(0, npts, size=max(1, npts // 10))] = 0.37
    infl = np.full(npts * nv, infl0)
    return dict(k=k, nv=nv, npts=npts, nens=nens, kld=kld, ensval=np.ascontiguousarray(ens), dep=dep, obs_off=off,
                obs_idx=idx, rdiag=rdiag, rloc=rloc, gues=gues, beta=beta, infl=infl, sp=1, sm=npts,
                sv=npts * nens, det_run=det_run)


def golden_case_list():
    """The covering case matrix of SURVEY.md section 8(c) for the letkf_core boundary.
    Each entry: dict(name, k, n, nobs, cond, infl, flags...) -- inputs come from core_case(seed)."""
    cases = []

    def add(k, n, cond="benign", infl=1.0, wloc=True, transm=True, pao=True, iu=False, det=False, pad=5):
        name = f"k{k}_n{n}_{cond}_i{infl}_w{wloc}_tm{int(transm)}_pa{int(pao)}_iu{iu}_d{int(det)}"
        cases.append(dict(name=name, k=k, n=n, nobs=max(n, 1) + pad, cond=cond, infl=infl, rdiag_wloc=wloc,
                          transm=transm, pao=pao, infl_update=iu, det=det,
                          seed=(k * 100003 + n * 101 + len(cases)) % (2 ** 31)))

    for k in (2, 3, 20, 50, 64, 100):
        for n in sorted({0, 1, k - 1, k, 200}):
            add(k, n, infl=(1.0, 1.1, 3.0)[(k + n) % 3], iu=bool((k + n) % 2), det=(n % 2 == 0))
    add(50, 1000, iu=True, det=True)
    add(50, 5000, iu=True)
    add(20, 5000, infl=1.1)
    add(100, 1000, det=True)
    # optional-argument combinations (common/common_letkf.f90:84-87, :188, :218-227)
    add(50, 200, wloc=False, iu=True, infl=1.1)
    add(50, 200, wloc=None, iu=None, transm=False, pao=False)   # all OPTIONALs absent: w-bar folded into trans
    add(50, 200, transm=False, pao=True)
    add(50, 200, transm=True, pao=False, det=True)
    add(20, 50, wloc=False, iu=False, transm=False, pao=True, infl=3.0)
    # conditioning
    for cond in ("ill", "dup", "zerocol"):
        add(50, 200, cond=cond, iu=True, det=True, infl=1.1)
        add(20, 37, cond=cond)
    # large-k paths (outputs stored as probes, see make_golden.py)
    add(320, 200, det=True)
    add(320, 1000, iu=True, infl=1.1)
    add(1000, 200)
    return cases


def golden_inputs(c):
    return core_case(c["k"], c["n"], c["seed"], nobs=c["nobs"], cond=c["cond"],
                     rdiag_wloc=bool(c["rdiag_wloc"]), infl=c["infl"], with_det=c["det"])


def probes(k):
    """Fixed probe vectors used to pin k x k outputs too large to store."""
    rng = np.random.default_rng(777 + k)
    return rng.standard_normal((k, 4))


def expected_status(c, inp):
    """LETKF_ST_* the library must report for a golden case: 3 when lambda_min < lambda_max * sqrt(eps)
    (common/common_mtx.f90:66-74), else 0; None when the spectrum sits within 1 % of that threshold."""
    n, k = c["n"], c["k"]
    if n == 0:
        return 0
    y = inp["hdxb"][:n]
    w = (1.0 / inp["rdiag"][:n]) if c["rdiag_wloc"] else inp["rloc"][:n] / inp["rdiag"][:n]
    shift = (k - 1) / c["infl"]
    # lambda(A) = shift + lambda(Y^T W Y); the non-zero part of the latter from the smaller Gram matrix
    yw = y * np.sqrt(w)[:, None]
    s = np.linalg.eigvalsh(yw.T @ yw if k <= n else yw @ yw.T)
    lmax = shift + max(s[-1], 0.0)
    lmin = shift + (max(s[0], 0.0) if k <= n else 0.0)
    ratio = lmin / lmax / 1.4901161193847656e-08
    if 0.99 < ratio < 1.01:
        return None
    return 3 if ratio < 1.0 else 0
