"""Synthetic SCALE-LETKF analysis workloads for bench.py, built on the device with torch.

Only plumbing lives here: random state, a radar-like observation lattice, and the per-point local-observation
lists that the reference's obs_local (scale/letkf/letkf_tools.f90:1325, no-limit mode :1438-1476) would hand to
letkf_core -- same cut-off tests and localisation weights as obs_local_cal (:1793-1906) with the float32-literal
cut-offs of letkf_obs.f90:27-28.  The hot path itself is only ever run through the C ABI.
"""
import math

import numpy as np
import torch

DIST_ZERO_FAC = float(np.float32(3.651483717))
DIST_ZERO_FAC_SQUARE = float(np.float32(13.33333333))

CONFIGS = {
    # BASELINE.json configs[1]: 240x240x60, k=50, ~200 local obs/point
    "C2": dict(nx=240, ny=240, nz=60, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=2900.0, err=3.0,
               ztop=18000.0, seed=20240609),
    # small stand-ins for tests / smoke-sized benches
    "C2-mini": dict(nx=48, ny=48, nz=12, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                    ztop=18000.0, seed=20240610),
    # production ensemble size (MEMBER=100 in 20 of the reference's 38 configs), small grid
    "C2-mini-k100": dict(nx=48, ny=48, nz=12, k=100, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                         ztop=18000.0, seed=20240611),
    "C2-mini-k20": dict(nx=48, ny=48, nz=12, k=20, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                        ztop=18000.0, seed=20240612),
    # the other register-kernel instantiations (KR = 16, 32, 48, 64) on a smaller grid: tests of the list-free and fused routes
    **{f"C2-tiny-k{k_}": dict(nx=24, ny=20, nz=7, k=k_, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                             ztop=18000.0, seed=20240620 + k_) for k_ in (16, 17, 30, 40, 51, 60)},
    # profiling stand-in: same grid, no observation in range (isolates state I/O + transform)
    "C2-mini-noobs": dict(nx=48, ny=48, nz=12, k=50, dx=1000.0, hloc=40.0, vloc=20.0, spacing=3200.0, err=3.0,
                          ztop=18000.0, seed=20240610),
    # BASELINE configs[2] (C3: k = 320) on a small grid: the large-k path (block Jacobi on the matrix cores)
    "C3-mini": dict(nx=24, ny=24, nz=6, k=320, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=2900.0, err=3.0,
                    ztop=18000.0, seed=20240613),
    # BASELINE configs[3] (C4: k = 50, radar-like dense obs, ~5000 local obs/point) on a small grid: the Gram dominates
    "C4-mini": dict(nx=32, ny=32, nz=8, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=600.0, err=3.0,
                    ztop=18000.0, seed=20240614),
    # Interior pieces of the full-size configurations: halo = True extends the observation lattice past the slab by
    # the localisation cut-off on every side, so that every point sees the local-observation count of the interior
    # of the full domain (C2's lattice: ~200; configs[3]: ~5000) instead of a boundary-thinned one.
    "C3-slab": dict(nx=24, ny=24, nz=6, k=320, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                    ztop=18000.0, seed=20240617, halo=True),
    "C4-slab": dict(nx=24, ny=24, nz=6, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=1100.0, err=3.0,
                    ztop=18000.0, seed=20240615, halo=True),
    "C5-slab": dict(nx=16, ny=16, nz=3, k=1000, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                    ztop=18000.0, seed=20240616, halo=True),
    "C2-slab-k100": dict(nx=48, ny=48, nz=12, k=100, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                         ztop=18000.0, seed=20240618, halo=True),
    # C2's geometry (60 levels, its observation lattice) at the production ensemble size, on a 48 x 48 piece of the domain
    "C2-cols-k100": dict(nx=48, ny=48, nz=60, k=100, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=2900.0, err=3.0,
                         ztop=18000.0, seed=20240619, halo=True),
    # SURVEY.md section 8(d)'s imbalance case: C2's grid and lattice, but observations only inside a radar disc of 60 km
    # radius around the domain centre (points further than the cut-off from the disc have no local observation at all)
    "C2-disc": dict(nx=240, ny=240, nz=60, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=2900.0, err=3.0,
                    ztop=18000.0, seed=20240620, disc=60000.0),
    "C2-mini-disc": dict(nx=48, ny=48, nz=12, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                         ztop=18000.0, seed=20240621, disc=12000.0),
    # few local observations per point (n ~ 20 < k): the rim of a radar disc
    "C2-mini-sparse": dict(nx=48, ny=48, nz=12, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=6000.0, err=3.0,
                           ztop=18000.0, seed=20240622, halo=True),
    # ---- the BASELINE configurations at their size (one GPU's share of the 8-GPU ones: a 4 x 2 tiling, the tile with
    # the localisation halo of observations around it).  Run with bench.py --level-slab (the lists of C4-gpu do not
    # fit at once: 10 M points x ~4900 x 20 B; the state of C5-gpu does not: 2 x 152 GB).
    "C3": dict(nx=240, ny=240, nz=60, k=320, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=2900.0, err=3.0,
               ztop=18000.0, seed=20240630),
    "C4-gpu": dict(nx=250, ny=500, nz=80, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=1100.0, err=3.0,
                   ztop=18000.0, seed=20240631, halo=True),
    # BASELINE configs[3] WHOLE: the 1000 x 1000 x 80 domain for `bench.py --gpus 8 --scaling strong --workload C4 --lists pipeline
    # --list-gb 24` (4 x 2 tiles of 250 x 500 x 80 = the C4-gpu tile, one per GPU; the state of the whole domain is 2 x 359 GB,
    # so there is no N = 1 run of it: its one-GPU number is C4-gpu's), and a small domain of the same density to rehearse the
    # tiling with
    "C4": dict(nx=1000, ny=1000, nz=80, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=1100.0, err=3.0,
               ztop=18000.0, seed=20240631),
    "C4-dom-mini": dict(nx=48, ny=32, nz=6, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=1100.0, err=3.0,
                        ztop=18000.0, seed=20240636),
    # configs[3]'s observation density at the ensemble size of 20 of the reference's 38 run configurations (MEMBER = 100), half of
    # the C4-gpu tile (the state of the whole tile at 101 slots would be 2 x 90 GB): meant for --max-nobs 100
    "C4h-k100": dict(nx=250, ny=250, nz=80, k=100, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=1100.0, err=3.0,
                     ztop=18000.0, seed=20240634, halo=True),
    "C5-gpu": dict(nx=120, ny=240, nz=60, k=1000, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=3200.0, err=3.0,
                   ztop=18000.0, seed=20240632, halo=True),
    # C2's grid at k = 20 (the memory-bound regime of configs[0]: arithmetic intensity ~5 flop/B)
    "C2-k20": dict(nx=240, ny=240, nz=60, k=20, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=2900.0, err=3.0,
                   ztop=18000.0, seed=20240633),
    # C2's grid at the small ensemble sizes of the reference's test configurations (MEMBER = 10 in 11, 8 in 9, 3 in 11 of its 58 config files)
    **{f"C2-k{k_}": dict(nx=240, ny=240, nz=60, k=k_, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=2900.0, err=3.0,
                        ztop=18000.0, seed=20240640 + k_) for k_ in (10, 8, 3)},
    # BASELINE configs[0] as SURVEY.md section 8(d) specifies it: 40 x 40 x 30 at DX = 15 km, k = 20, 500 observations at
    # uniformly random positions, conventional upper-air type (ADPUPA): vertical localisation in ln p (letkf_tools.f90:1864,
    # scale 0.4), HORI_LOCAL 500 km -- the horizontal cut-off covers the whole domain, so n is whatever the ln-p test leaves of
    # the 500 (round 3's C1 was a 400-observation radar lattice with z localisation)
    "C1": dict(nx=40, ny=40, nz=30, k=20, dx=15000.0, hloc=500000.0, vloc=0.4, err=1.0, ztop=18000.0, seed=20240608,
               obs="random", nobs=500, vmode=0),
    # configs[3]'s observation density inside a radar disc, nothing outside: columns with ~16 k, a few hundred and no horizontal
    # survivors in one call (the list-free route's batches, passes and empty columns)
    "C4-slab-disc": dict(nx=48, ny=44, nz=7, k=50, dx=1000.0, hloc=4000.0, vloc=2000.0, spacing=1100.0, err=3.0,
                         ztop=18000.0, seed=20240635, halo=True, disc=16000.0),
}


def level_heights(nz, ztop):
    """stretched levels 50 m .. ztop (flat terrain)"""
    s = np.linspace(0.0, 1.0, nz)
    return 50.0 + (ztop - 50.0) * s ** 1.6


def lattice(cfg, device):
    """Observation lattice coordinates (metres) and the number of halo planes on the low side (0 without halo)."""
    sp_o, f64 = cfg["spacing"], torch.float64
    nh = nhz = 0
    if cfg.get("halo"):
        nh = int(math.ceil(cfg["hloc"] * DIST_ZERO_FAC / sp_o))
        nhz = int(math.ceil(cfg["vloc"] * DIST_ZERO_FAC / sp_o))
    ax = lambda length, n: (torch.arange(-n, int(math.ceil(length / sp_o - 0.5)) + n, device=device, dtype=f64) + 0.5) * sp_o
    return ax(cfg["nx"] * cfg["dx"], nh), ax(cfg["ny"] * cfg["dx"], nh), ax(cfg["ztop"], nhz), nh, nhz


def disc_mask(cfg, ox, oy):
    """[noy, nox] mask of the lattice columns that carry observations: all of them, or (cfg["disc"] = radius in metres)
    those inside a radar disc around the domain centre."""
    if not cfg.get("disc"):
        return torch.ones(len(oy), len(ox), dtype=torch.bool, device=ox.device)
    cx, cy = 0.5 * cfg["nx"] * cfg["dx"], 0.5 * cfg["ny"] * cfg["dx"]
    return ((ox[None, :] - cx) ** 2 + (oy[:, None] - cy) ** 2) <= cfg["disc"] ** 2


def build(cfg_name, device, nv=11, det_run=False, rank=0, world=1, ensval_kind="iid", lists=True):
    """lists = False: no torch-built local-observation lists (the full-size configurations: the lists come from the
    device search, one level slab at a time)."""
    cfg = CONFIGS[cfg_name]
    if cfg.get("obs") == "random":
        return build_random(cfg_name, device, nv=nv, det_run=det_run, rank=rank, lists=lists)
    nx, ny, nz, k = cfg["nx"], cfg["ny"], cfg["nz"], cfg["k"]
    dx, hloc, vloc, sp_o, err = cfg["dx"], cfg["hloc"], cfg["vloc"], cfg["spacing"], cfg["err"]
    g = torch.Generator(device=device)
    g.manual_seed(cfg["seed"] + 7919 * rank)
    f64 = torch.float64
    # ---- observation lattice (type-22 radar-like: vertical localisation in z, letkf_tools.f90:1857)
    ox, oy, oz, nh, nhz = lattice(cfg, device)
    nox, noy, noz = len(ox), len(oy), len(oz)
    nobs = nox * noy * noz
    kld = k + 1
    dmask = disc_mask(cfg, ox, oy)
    ensval = torch.randn(nobs, kld, generator=g, device=device, dtype=f64) * 2.0
    ensval[:, :k] -= ensval[:, :k].mean(dim=1, keepdim=True)
    dep = torch.randn(nobs, generator=g, device=device, dtype=f64) * math.sqrt(err * err + 4.0)
    # ---- per-point local lists, one level slab at a time.  Point index pt = ij + nij*ilev (point-fastest, as
    #      gues3d(nij1,nlev,...)); obs index = (iz*noy + iy)*nox + ix.
    zlev = torch.from_numpy(level_heights(nz, cfg["ztop"])).to(device)
    gx = (torch.arange(nx, device=device, dtype=f64) + 0.5) * dx
    gy = (torch.arange(ny, device=device, dtype=f64) + 0.5) * dx
    px = gx.repeat(ny)                      # ij = i + nx*j
    py = gy.repeat_interleave(nx)
    rh = int(math.ceil(hloc * DIST_ZERO_FAC / sp_o))
    rv = int(math.ceil(vloc * DIST_ZERO_FAC / sp_o))
    offs = torch.arange(-rh, rh + 1, device=device)
    offv = torch.arange(-rv, rv + 1, device=device)
    nij = nx * ny
    cix = torch.floor(px / sp_o).long() + nh
    ciy = torch.floor(py / sp_o).long() + nh
    counts_all, idx_all, rloc_all = [], [], []
    for lev in range(nz if lists else 0):
        ciz = int(math.floor(float(zlev[lev]) / sp_o)) + nhz
        ix = cix[:, None] + offs[None, :]                                   # [nij, nh]
        iy = ciy[:, None] + offs[None, :]
        iz = ciz + offv                                                     # [nvv]
        okx = (ix >= 0) & (ix < nox)
        oky = (iy >= 0) & (iy < noy)
        okz = (iz >= 0) & (iz < noz)
        ddx = (px[:, None] - ox[ix.clamp(0, nox - 1)])                      # metres
        ddy = (py[:, None] - oy[iy.clamp(0, noy - 1)])
        ndv = (zlev[lev] - oz[iz.clamp(0, noz - 1)]).abs() / vloc           # [nvv]
        # candidate order: z, then y, then x (ascending obs index)
        d2h = (ddy[:, None, :, None] ** 2 + ddx[:, None, None, :] ** 2)     # [nij,1,nh,nh]
        ndh = torch.sqrt(d2h) / hloc
        nd2 = ndh * ndh + (ndv * ndv)[None, :, None, None]
        ok = (okz & (ndv <= DIST_ZERO_FAC))[None, :, None, None] & oky[:, None, :, None] & okx[:, None, None, :]
        ok = ok & (ndh <= DIST_ZERO_FAC) & (nd2 <= DIST_ZERO_FAC_SQUARE)
        if cfg.get("disc"):
            ok = ok & dmask[iy.clamp(0, noy - 1)[:, None, :, None], ix.clamp(0, nox - 1)[:, None, None, :]]
        oidx = (iz[None, :, None, None] * noy + iy[:, None, :, None]) * nox + ix[:, None, None, :]
        okf = ok.reshape(nij, -1)
        counts_all.append(okf.sum(dim=1))
        sel = okf.reshape(-1).nonzero(as_tuple=False).squeeze(1)
        idx_all.append(oidx.reshape(-1)[sel].to(torch.int32))
        rloc_all.append(torch.exp(-0.5 * nd2.reshape(-1)[sel]))
        del d2h, ndh, nd2, ok, oidx, okf, sel
    if lists:
        counts = torch.cat(counts_all)
        obs_off = torch.zeros(nij * nz + 1, dtype=torch.int64, device=device)
        obs_off[1:] = torch.cumsum(counts, 0)
        obs_idx = torch.cat(idx_all)
        rloc = torch.cat(rloc_all)
        rdiag = (err * err) / rloc                                          # letkf_tools.f90:1903
    else:
        counts = torch.zeros(1, dtype=torch.int64, device=device)
        obs_off = obs_idx = rloc = rdiag = None
    del counts_all, idx_all, rloc_all
    # ---- ensemble state gues3d(nij*nz, nens, nv): members, then mean slot k, det slot k+1
    npts = nij * nz
    nens = k + 1 + (1 if det_run else 0)
    gues = torch.empty(nv * nens * npts, dtype=f64, device=device)
    gv = gues.view(nv, nens, npts)
    sig = [2.0, 2.0, 2.0, 1.0, 50.0] + [1e-3] * (nv - 5)
    mean0 = [10.0, 5.0, 0.1, 280.0, 8.0e4] + [5e-3] * (nv - 5)
    for v in range(nv):
        gv[v].normal_(mean0[v], sig[v], generator=g)
    if ensval_kind == "correlated":
        # spatially smooth member perturbations (coarse noise, trilinear upsampling) under the white part, so that the
        # H-like obs-space perturbations of correlate_ensval() are correlated between neighbouring observations
        import torch.nn.functional as F
        nzc, nyc, nxc = max(2, nz // 4), max(2, ny // 8), max(2, nx // 8)
        for v in range(nv):
            coarse = torch.randn(1, k, nzc, nyc, nxc, generator=g, device=device, dtype=torch.float32)
            up = F.interpolate(coarse, size=(nz, ny, nx), mode="trilinear", align_corners=True)[0].reshape(k, npts)
            gv[v, :k] = mean0[v] + 0.3 * (gv[v, :k] - mean0[v]) + (0.95 * sig[v]) * up.to(f64)
            del coarse, up
    return dict(cfg=cfg, name=cfg_name, k=k, nv=nv, npts=npts, nens=nens, kld=kld, nobs=nobs, ensval=ensval, dep=dep,
                obs_off=obs_off, obs_idx=obs_idx, rdiag=rdiag, rloc=rloc, gues=gues, sp=1, sm=npts, sv=npts * nens,
                n_mean=float(counts.double().mean()), n_max=int(counts.max()), det_run=det_run, ensval_kind=ensval_kind,
                sig=sig, gen=g)


def p_mean(z):
    """p = 1000 hPa * exp(-z / 7.5 km)"""
    return 1.0e5 * torch.exp(-z / 7500.0)


def build_random(cfg_name, device, nv=11, det_run=False, rank=0, lists=True):
    """Workloads whose observations sit at uniformly random positions (cfg["obs"] == "random"; C1): conventional type,
    vertical localisation in ln p (vmode 0: |ln p_obs - ln p_point| / vloc, letkf_tools.f90:1864).  The lists are the
    brute-force evaluation of obs_local_cal (:1793-1906) for every (point, observation) pair in table order."""
    cfg = CONFIGS[cfg_name]
    nx, ny, nz, k, dx = cfg["nx"], cfg["ny"], cfg["nz"], cfg["k"], cfg["dx"]
    hloc, vloc, err, nobs = cfg["hloc"], cfg["vloc"], cfg["err"], cfg["nobs"]
    f64 = torch.float64
    g = torch.Generator(device=device)
    g.manual_seed(cfg["seed"] + 7919 * rank)
    u = torch.rand(3, nobs, generator=g, device=device, dtype=f64)
    ob_x, ob_y, ob_z = u[0] * nx * dx, u[1] * ny * dx, u[2] * cfg["ztop"]
    ob_p = p_mean(ob_z)
    kld = k + 1
    ensval = torch.randn(nobs, kld, generator=g, device=device, dtype=f64) * 2.0
    ensval[:, :k] -= ensval[:, :k].mean(dim=1, keepdim=True)
    dep = torch.randn(nobs, generator=g, device=device, dtype=f64) * math.sqrt(err * err + 4.0)
    zlev = torch.from_numpy(level_heights(nz, cfg["ztop"])).to(device)
    plev = p_mean(zlev)
    nij = nx * ny
    npts = nij * nz
    # the table order of search_tables(): rows sorted by mesh cell (j, i), stable -- the brute-force lists below follow the
    # candidate order of obs_local (mesh row j, then table row), which for a rectangle that covers whole mesh rows IS table order
    counts = torch.zeros(1, dtype=torch.int64, device=device)
    obs_off = obs_idx = rloc = rdiag = None
    if lists:
        _, order = random_mesh(cfg, ob_x, ob_y)
        sx, sy, sp_ = ob_x[order] / dx, ob_y[order] / dx, ob_p[order]
        gx = (torch.arange(nx, device=device, dtype=f64) + 0.5)
        gy = (torch.arange(ny, device=device, dtype=f64) + 0.5)
        pri, prj = gx.repeat(ny), gy.repeat_interleave(nx)
        rdx = (pri[:, None] - sx[None, :]) * dx
        rdy = (prj[:, None] - sy[None, :]) * dx
        ndh = torch.sqrt(rdx * rdx + rdy * rdy) / hloc                         # [nij, nobs]
        cnt_l, idx_l, rl_l = [], [], []
        for lev in range(nz):
            ndv = (torch.log(sp_) - torch.log(plev[lev])).abs() / vloc           # [nobs]
            nd2 = ndh * ndh + (ndv * ndv)[None, :]
            ok = (ndv <= DIST_ZERO_FAC)[None, :] & (ndh <= DIST_ZERO_FAC) & (nd2 <= DIST_ZERO_FAC_SQUARE)
            cnt_l.append(ok.sum(dim=1))
            sel = ok.reshape(-1).nonzero(as_tuple=False).squeeze(1)
            idx_l.append(order[sel % nobs].to(torch.int32))                      # (entries name ORIGINAL rows, as build())
            rl_l.append(torch.exp(-0.5 * nd2.reshape(-1)[sel]))
        counts = torch.cat(cnt_l)
        obs_off = torch.zeros(npts + 1, dtype=torch.int64, device=device)
        obs_off[1:] = torch.cumsum(counts, 0)
        obs_idx, rloc = torch.cat(idx_l), torch.cat(rl_l)
        rdiag = (err * err) / rloc
    nens = k + 1 + (1 if det_run else 0)
    gues = torch.empty(nv * nens * npts, dtype=f64, device=device)
    gv = gues.view(nv, nens, npts)
    sig = [2.0, 2.0, 2.0, 1.0, 50.0] + [1e-3] * (nv - 5)
    mean0 = [10.0, 5.0, 0.1, 280.0, 8.0e4] + [5e-3] * (nv - 5)
    for v in range(nv):
        gv[v].normal_(mean0[v], sig[v], generator=g)
    return dict(cfg=cfg, name=cfg_name, k=k, nv=nv, npts=npts, nens=nens, kld=kld, nobs=nobs, ensval=ensval, dep=dep,
                obs_off=obs_off, obs_idx=obs_idx, rdiag=rdiag, rloc=rloc, gues=gues, sp=1, sm=npts, sv=npts * nens,
                n_mean=float(counts.double().mean()), n_max=int(counts.max()), det_run=det_run, ensval_kind="iid",
                sig=sig, gen=g, ob_x=ob_x, ob_y=ob_y, ob_lev=ob_p)


def mesh_shape(cfg):
    """The sorting mesh of set_letkf_obs for one combined type (letkf_obs.f90:655-695): (ngrd_i, ngrd_j, nsch_i, nsch_j)."""
    nx, ny, dx, hloc = cfg["nx"], cfg["ny"], cfg["dx"], cfg["hloc"]
    spc = hloc * DIST_ZERO_FAC / 6.0
    ngrd_i = min(math.ceil(dx * nx / spc), nx)
    ngrd_j = min(math.ceil(dx * ny / spc), ny)
    nsch_i = math.ceil(hloc * DIST_ZERO_FAC / (dx * nx / ngrd_i))
    nsch_j = math.ceil(hloc * DIST_ZERO_FAC / (dx * ny / ngrd_j))
    return ngrd_i, ngrd_j, nsch_i, nsch_j


def random_mesh(cfg, ob_x, ob_y):
    """(cell index per observation, stable sort order by cell) on that mesh, for observations at (ob_x, ob_y) metres."""
    nx, ny, dx = cfg["nx"], cfg["ny"], cfg["dx"]
    ngrd_i, ngrd_j, nsch_i, nsch_j = mesh_shape(cfg)
    next_i, next_j = ngrd_i + 2 * nsch_i, ngrd_j + 2 * nsch_j
    ogi = (torch.ceil(ob_x / dx * ngrd_i / nx).long() + nsch_i).clamp(1, next_i)
    ogj = (torch.ceil(ob_y / dx * ngrd_j / ny).long() + nsch_j).clamp(1, next_j)
    cell = (ogj - 1) * next_i + (ogi - 1)
    return cell, torch.argsort(cell, stable=True)


def state_view(w, t):
    """[v, m, p] view of a state buffer in the workload's layout (element (p, m, v) at p*sp + m*sm + v*sv)."""
    return torch.as_strided(t, (w["nv"], w["nens"], w["npts"]), (w["sv"], w["sm"], w["sp"]))


def relayout_state(w, layout):
    """layout "member": the point-major, member-fastest layout the C ABI's strides also allow (sm = 1, sv = nens,
    sp = nens*nv: a point's 11 x (k+1) doubles are one contiguous 4.5 KB block) instead of the reference's
    gues3d(nij1*nlev, nens, nv3d) (sp = 1)."""
    if layout != "member":
        return
    nv, nens, npts = w["nv"], w["nens"], w["npts"]
    g = w["gues"].view(nv, nens, npts).permute(2, 0, 1).contiguous()      # [p][v][m]
    w["gues"] = g.reshape(-1)
    w["sp"], w["sm"], w["sv"] = nens * nv, 1, nens
    del g


def correlate_ensval(w):
    """obs-space perturbations as an H-like linear combination of the state perturbations
    around the observation (here: of u and T at the nearest grid point) plus unit noise, so that Y and X' -- and the
    rows of Y among themselves -- are correlated.  Call after the perturbation pass (gues holds x')."""
    cfg = w["cfg"]
    dev_ = w["gues"].device
    nx, ny, nz, k = cfg["nx"], cfg["ny"], cfg["nz"], w["k"]
    ox, oy, oz, _, _ = lattice(cfg, dev_)
    nox, noy = len(ox), len(oy)
    zlev = torch.from_numpy(level_heights(nz, cfg["ztop"])).to(dev_)
    ix = torch.clamp(torch.floor(ox / cfg["dx"]).long(), 0, nx - 1)
    iy = torch.clamp(torch.floor(oy / cfg["dx"]).long(), 0, ny - 1)
    iz = torch.argmin((oz[:, None] - zlev[None, :]).abs(), dim=1)
    # lattice row index = (iz*noy + iy)*nox + ix
    pt = ((iz[:, None, None] * ny + iy[None, :, None]) * nx + ix[None, None, :]).reshape(-1)
    gv = state_view(w, w["gues"])
    sig = w["sig"]
    noise = torch.randn(w["nobs"], k, generator=w["gen"], device=dev_, dtype=torch.float64)
    noise -= noise.mean(dim=1, keepdim=True)
    y = (1.2 / sig[0]) * gv[0, :k][:, pt].T + (0.8 / sig[3]) * gv[3, :k][:, pt].T + 0.5 * noise
    w["ensval"][:, :k] = 1.6 * y
    del noise, y


def alg_bytes_per_solve(n, k, nv, det=False):
    """every solve streams its own local-obs slice once; no k x k matrix touches HBM."""
    return 8.0 * (n * k + 3 * n + (n if det else 0) + k * nv + nv + k * nv + 1)


def alg_flops_per_solve(n, k, nv, rtps=True):
    """SURVEY.md section 8(d), the k x k ("primal") count: Gram, Y^T d, symmetric eigensolve with vectors (nominal
    9 k^3), T, w-bar, transform (+ Pa for RTPS)."""
    f = 2.0 * n * k * k + 2.0 * n * k + 9.0 * k ** 3 + 2.0 * k ** 3 + 2.0 * k * k + 2.0 * nv * k * k
    if rtps:
        f += 2.0 * k ** 3 + 2.0 * nv * k * k
    return f


def alg_flops_dual(n, k, nv):
    """The same analysis through the n x n observation-space eigenproblem (n < k; letkf_staged.hip): Z Z^T, the
    eigensolve at order n (nominal 9 n^3), and for the nb = nv + 2 right-hand sides Z b, U^T t, U c, Z^T q.  What a
    point with fewer observations than members REQUIRES is the smaller of the two counts."""
    nb = nv + 2
    return 2.0 * n * n * k + 9.0 * n ** 3 + 2.0 * (2.0 * n * k * nb) + 2.0 * (2.0 * n * n * nb) + 4.0 * nv * k


def alg_flops_poly(n, k, nv, deg):
    """The analysis without an eigen-decomposition (letkf_krylov.hip): the Gram of the smaller formulation (order
    m = min(n, k)), deg products of the m x m matrix with the nb = nv + 2 right-hand sides (CG iterations), the CG vector
    updates and the combination of the residual history, Z b and Z^T q."""
    nb = nv + 2
    m = min(n, k)
    side = 2.0 * (2.0 * n * k * nb) if n < k else 2.0 * n * k * 2
    return 2.0 * m * m * max(n, k) + deg * 2.0 * m * m * nb + side + 14.0 * deg * m * nb + 4.0 * nv * k


def alg_flops_required(n, k, nv, rtps=True):
    f = alg_flops_per_solve(n, k, nv, rtps)
    return min(f, alg_flops_dual(n, k, nv)) if n < k else f


def search_tables(w, pkg, device, max_nobs=0):
    """The same C2-style lattice as build(), described the way set_letkf_obs leaves it behind (one radar ctype, mesh of
    letkf_obs.f90:655-695, rows sorted (j, i)), so that the lists can come from letkf_obs_search_dev instead of
    torch.  Returns (tables struct, keepalive, row order, point coordinate tensors)."""
    cfg = w["cfg"]
    nx, ny, nz = cfg["nx"], cfg["ny"], cfg["nz"]
    dx, hloc, vloc = cfg["dx"], cfg["hloc"], cfg["vloc"]
    f64 = torch.float64
    random_obs = cfg.get("obs") == "random"
    if random_obs:
        ri, rj, lev = w["ob_x"] / dx, w["ob_y"] / dx, w["ob_lev"]
        nlat = ri.numel()
        kept = torch.arange(nlat, device=device)
    else:
        sp_o = cfg["spacing"]
        ox, oy, oz, _, _ = lattice(cfg, device)
        nox, noy, noz = len(ox), len(oy), len(oz)
        # lattice row index = (iz*noy + iy)*nox + ix  (as in build())
        ri = (ox / dx).repeat(noy * noz)
        rj = (oy / dx).repeat_interleave(nox).repeat(noz)
        lev = oz.repeat_interleave(nox * noy)
        nlat = ri.numel()                                  # rows of the full lattice (= rows of w["ensval"])
        kept = disc_mask(cfg, ox, oy).reshape(-1).repeat(noz).nonzero(as_tuple=False).squeeze(1)   # lattice rows with an observation
        ri, rj, lev = ri[kept], rj[kept], lev[kept]
    ngrd_i, ngrd_j, nsch_i, nsch_j = mesh_shape(cfg)
    next_i, next_j = ngrd_i + 2 * nsch_i, ngrd_j + 2 * nsch_j
    # (lattice rows beyond the extended mesh -- a halo lattice can reach a fraction of a spacing past the cut-off -- are
    # clamped into its edge cells: they are outside every point's cut-off anyway)
    ogi = (torch.ceil(ri * ngrd_i / nx).long() + nsch_i).clamp(1, next_i)
    ogj = (torch.ceil(rj * ngrd_j / ny).long() + nsch_j).clamp(1, next_j)
    cell = (ogj - 1) * next_i + (ogi - 1)
    order = torch.argsort(cell, stable=True)
    ri, rj, lev = ri[order], rj[order], lev[order]
    order = kept[order]                                # table row -> lattice row
    counts = torch.bincount(cell, minlength=next_i * next_j).view(next_j, next_i)
    ends = torch.cumsum(counts.reshape(-1), 0).view(next_j, next_i)
    ac = torch.zeros(next_j, next_i + 1, dtype=torch.int64, device=device)
    ac[:, 1:] = ends
    ac[1:, 0] = ends[:-1, -1]
    t = pkg.SearchTables()
    t.nctype, t.ngroup, t.criterion, t.nlon, t.nlat = 1, 1, 1, nx, ny
    t.dx, t.dy, t.i_org, t.j_org, t.rain_base = dx, dx, 0.0, 0.0, 8.5e4
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    d64 = lambda v: torch.tensor(v, dtype=f64, device=device)
    keep = dict(group_start=i32([0, 1]), group_member=i32([0]), vmode=i32([cfg.get("vmode", 1)]), hori_loc=d64([hloc]),
                vert_loc=d64([vloc]), varloc=d64([1.0]), max_nobs=i32([0]), ngrd_i=i32([ngrd_i]), ngrd_j=i32([ngrd_j]),
                ngrdsch_i=i32([nsch_i]), ngrdsch_j=i32([nsch_j]), ngrdext_i=i32([next_i]), ngrdext_j=i32([next_j]),
                ac_off=torch.zeros(1, dtype=torch.int64, device=device), ac_ext=ac.reshape(-1).to(torch.int32),
                ob_ri=ri.contiguous(), ob_rj=rj.contiguous(), ob_lev=lev.contiguous(),
                ob_dat=torch.full_like(ri, 1.0e5), ob_err=torch.full_like(ri, cfg["err"]))
    t.limit_hint = 2 if max_nobs > 0 else 1
    if max_nobs > 0:
        # SURVEY.md section 8(d) variant: two radar ctypes (REF, Vr) on the same lattice, each limited to max_nobs
        # observations per grid point.  The second ctype's rows follow the first's in the table (its own prefix sums
        # shifted by the row count) and get their own obs-space perturbations and departures.
        nrow = ri.numel()
        g2 = torch.Generator(device=device)
        g2.manual_seed(cfg["seed"] + 4242)
        ens2 = torch.randn(w["ensval"].shape, generator=g2, device=device, dtype=f64) * 2.0
        ens2[:, :w["k"]] -= ens2[:, :w["k"]].mean(dim=1, keepdim=True)
        dep2 = torch.randn(nlat, generator=g2, device=device, dtype=f64) * float(w["dep"].std())
        w["ensval"] = torch.cat([w["ensval"], ens2])
        w["dep"] = torch.cat([w["dep"], dep2])
        order = torch.cat([order, order + nlat])
        t.nctype, t.ngroup = 2, 2
        rep = lambda a: a.repeat(2)
        keep = dict(group_start=i32([0, 1, 2]), group_member=i32([0, 1]), vmode=i32([1, 1]),
                    hori_loc=rep(keep["hori_loc"]), vert_loc=rep(keep["vert_loc"]), varloc=d64([1.0, 1.0]),
                    max_nobs=i32([max_nobs, max_nobs]), ngrd_i=rep(keep["ngrd_i"]), ngrd_j=rep(keep["ngrd_j"]),
                    ngrdsch_i=rep(keep["ngrdsch_i"]), ngrdsch_j=rep(keep["ngrdsch_j"]),
                    ngrdext_i=rep(keep["ngrdext_i"]), ngrdext_j=rep(keep["ngrdext_j"]),
                    ac_off=torch.tensor([0, keep["ac_ext"].numel()], dtype=torch.int64, device=device),
                    ac_ext=torch.cat([keep["ac_ext"], keep["ac_ext"] + nrow]).contiguous(),
                    ob_ri=rep(keep["ob_ri"]), ob_rj=rep(keep["ob_rj"]), ob_lev=rep(keep["ob_lev"]),
                    ob_dat=rep(keep["ob_dat"]), ob_err=rep(keep["ob_err"]))
    for k, v in keep.items():
        setattr(t, k, v.data_ptr())
    zlev = torch.from_numpy(level_heights(nz, cfg["ztop"])).to(device)
    gx = (torch.arange(nx, device=device, dtype=f64) + 0.5)
    gy = (torch.arange(ny, device=device, dtype=f64) + 0.5)
    pri = gx.repeat(ny).repeat(nz)
    prj = gy.repeat_interleave(nx).repeat(nz)
    prz = zlev.repeat_interleave(nx * ny)
    prl = p_mean(prz)                                  # (only read by the ln-p vertical modes)
    return t, keep, order, (pri, prj, prl, prz)


def remap_lists_to_sorted(obs_idx, order, n_rows_unsorted):
    """The torch builder's list entries index the observation table in its original order; `order` (search_tables) is the
    row order of the sorted table (sorted row i = original row order[i], possibly a subset, possibly with repeats shifted by
    the table length for a second combined type).  Returns (the entries as rows of the SORTED table, whether every entry
    has one): an entry that names a row the sorted table lost maps to row 0 and makes the second value False."""
    inv = torch.full((n_rows_unsorted,), -1, dtype=torch.int64, device=order.device)
    inv[order.long()] = torch.arange(order.numel(), device=order.device, dtype=torch.int64)
    mapped = inv[obs_idx.long()]
    ok = bool((mapped >= 0).all().item())
    return mapped.clamp(min=0).to(obs_idx.dtype), ok


def sample_points(w, pts):
    """Host copies of everything the loop body reads for the grid points `pts` (sorted numpy int64): CSR lists
    re-based to the sample, the sample's slice of the state.  For the CPU checker / baseline (bench.py, tests)."""
    pts = np.asarray(pts, dtype=np.int64)
    ns = len(pts)
    dev = w["obs_off"].device
    tp = torch.from_numpy(pts).to(dev)
    o0 = w["obs_off"][tp]
    cnt = (w["obs_off"][tp + 1] - o0)
    off = torch.zeros(ns + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(cnt, 0)
    nnz = int(off[-1].item())
    # entry e of the sample list belongs to sample point searchsorted(off, e, right) - 1
    e = torch.arange(nnz, dtype=torch.int64, device=dev)
    sp_ = torch.searchsorted(off, e, right=True) - 1
    sel = o0[sp_] + (e - off[sp_])
    nv, nens, npts = w["nv"], w["nens"], w["npts"]
    gv = state_view(w, w["gues"])[:, :, tp].contiguous().cpu().numpy().reshape(-1)
    return dict(off=off.cpu().numpy(), idx=w["obs_idx"][sel].cpu().numpy(), rdiag=w["rdiag"][sel].cpu().numpy(),
                rloc=w["rloc"][sel].cpu().numpy(), gues=gv, ns=ns, pts=pts)
