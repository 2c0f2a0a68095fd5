"""bench.py --scaling strong: ONE analysis domain cut into px x py tiles, one rank (= GPU = SCALE subdomain) per tile.

Every timed step runs, per rank, the whole path of a subdomain rank of the reference:
  set_letkf_obs   bucket sort of the rank's own observations onto the per-ctype mesh        letkf_obs_mesh_sort_dev
                  ALLGATHERV of the sorted buffers + cell counts over the ranks (the path's    letkf_obs_allgatherv_dev /
                  one exchange, scale/letkf/letkf_obs.f90:826-831, 1036-1046)                  torch.distributed
                  extended-subdomain plan with the localisation halo (:922-976, 1059-1109)    letkf_obs_halo_plan_dev
                  copy into obsda_sort                                                         letkf_obs_gather_rows_dev
  das_letkf       obs_local for the rank's grid points                                         letkf_obs_search_columns_dev
                  the loop body                                                                letkf_das_points_dev
(the departure / QC stage is left out: the synthetic table already holds perturbations and departures).  value = all
grid points of the domain / (max over ranks of the step time): total work is fixed as N grows.  That the tiled analysis
equals the single-domain one is what tests/test_gpu_tiles.py proves with virtual ranks; here the ranks are real.
Plumbing only (torch for memory and the process group); every compute call goes through the C ABI."""
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

import bench_workload as bw


def run(args, ctx, pkg, dev, rank, world):
    import importlib
    sharding = importlib.import_module("scale_letkf_amd.sharding")
    cfg = bw.CONFIGS[args.workload]
    assert not cfg.get("halo"), "--scaling strong takes the domain workloads (C2, C2-mini, ...)"
    nx, ny, nz, k = cfg["nx"], cfg["ny"], cfg["nz"], cfg["k"]
    dx, hloc, vloc, err = cfg["dx"], cfg["hloc"], cfg["vloc"], cfg["err"]
    px, py = sharding.tile_grid(world)
    assert nx % px == 0 and ny % py == 0, "tiles must divide the domain"
    nlon, nlat = nx // px, ny // py
    pi, pj = rank % px, rank // px
    f64 = torch.float64
    nv, kld = 11, k + 1
    # ---- the global observation lattice (same on every rank: one seed), and this rank's share of it
    ox, oy, oz, _, _ = bw.lattice(cfg, dev)
    nox, noy, noz = len(ox), len(oy), len(oz)
    nobs = nox * noy * noz
    g = torch.Generator(device=dev)
    g.manual_seed(cfg["seed"])
    ens_g = torch.randn(nobs, kld, generator=g, device=dev, dtype=f64) * 2.0
    ens_g[:, :k] -= ens_g[:, :k].mean(dim=1, keepdim=True)
    dep_g = torch.randn(nobs, generator=g, device=dev, dtype=f64) * math.sqrt(err * err + 4.0)
    ri_g = (ox / dx).repeat(noy * noz) + 0.5                       # ri - 0.5 = metric position in grid units (IHALO = 0)
    rj_g = (oy / dx).repeat_interleave(nox).repeat(noz) + 0.5
    lev_g = oz.repeat_interleave(nox * noy)
    mine = ((ri_g - 0.5 > pi * nlon) & (ri_g - 0.5 <= (pi + 1) * nlon) &
            (rj_g - 0.5 > pj * nlat) & (rj_g - 0.5 <= (pj + 1) * nlat)).nonzero(as_tuple=False).squeeze(1)
    loc = torch.cat([ens_g[mine], dep_g[mine, None], ri_g[mine, None], rj_g[mine, None], lev_g[mine, None]], dim=1).contiguous()
    ncols = kld + 4
    nloc = loc.shape[0]
    ctype = torch.zeros(nloc, dtype=torch.int32, device=dev)
    qc = torch.zeros(nloc, dtype=torch.int32, device=dev)
    ri_l, rj_l = loc[:, kld + 1].contiguous(), loc[:, kld + 2].contiguous()
    del ens_g, dep_g
    # ---- sorting mesh of the subdomain (letkf_obs.f90:655-695)
    spc = hloc * bw.DIST_ZERO_FAC / 6.0
    ngi, ngj = min(math.ceil(dx * nlon / spc), nlon), min(math.ceil(dx * nlat / spc), nlat)
    nsi = math.ceil(hloc * bw.DIST_ZERO_FAC / (dx * nlon / ngi))
    nsj = math.ceil(hloc * bw.DIST_ZERO_FAC / (dx * nlat / ngj))
    h32 = lambda v: np.array([v], dtype=np.int32)
    gi, gj, si, sj = h32(ngi), h32(ngj), h32(nsi), h32(nsj)
    mesh = pkg.Mesh()
    mesh.nctype, mesh.nlon, mesh.nlat, mesh.ihalo, mesh.jhalo, mesh.rank_i, mesh.rank_j = 1, nlon, nlat, 0, 0, pi, pj
    mesh.ngrd_i, mesh.ngrd_j = gi.ctypes.data, gj.ctypes.data
    # non-square tiles (N = 2, 8 on a square domain): sort with ngrd_j like the lookup does, or the reference's
    # ij_obsgrd (letkf_obs.f90:1200) loses observations and the tilings would not do the same work
    mesh.fix_ij_obsgrd = 1
    lay = pkg.HaloLayout()
    lay.nctype, lay.nprocs, lay.prc_num_x, lay.myrank = 1, world, px, rank
    lay.ngrd_i, lay.ngrd_j, lay.ngrdsch_i, lay.ngrdsch_j = gi.ctypes.data, gj.ctypes.data, si.ctypes.data, sj.ctypes.data
    ncell = ngi * ngj
    nacx = (ngi + 2 * nsi + 1) * (ngj + 2 * nsj)
    # ---- this rank's grid points p = ij + nij1*lev and their ensemble (random: the timing does not depend on it)
    nij1, npts = nlon * nlat, nlon * nlat * nz
    zlev = torch.from_numpy(bw.level_heights(nz, cfg["ztop"])).to(dev)
    ii = torch.arange(nlon, device=dev, dtype=f64) + pi * nlon + 1.0
    jj = torch.arange(nlat, device=dev, dtype=f64) + pj * nlat + 1.0
    rig, rjg = ii.repeat(nlat).contiguous(), jj.repeat_interleave(nlon).contiguous()
    prz = zlev.repeat_interleave(nij1).contiguous()
    prl = torch.full_like(prz, 1.0e5)
    nens = k + 1
    gs = torch.Generator(device=dev)
    gs.manual_seed(cfg["seed"] + 7919 * (rank + 1))
    gues = torch.empty(nv * nens * npts, dtype=f64, device=dev)
    gv = gues.view(nv, nens, npts)
    sig = [2.0, 2.0, 2.0, 1.0, 50.0] + [1e-3] * (nv - 5)
    mean0 = [10.0, 5.0, 0.1, 280.0, 8.0e4] + [5e-3] * (nv - 5)
    for v in range(nv):
        gv[v].normal_(mean0[v], sig[v], generator=gs)
    ctx.ens_mean(k, nv, npts, gues, 1, npts, npts * nens)
    ctx.to_perturbations(k, nv, npts, gues, 1, npts, npts * nens)
    anal = torch.empty_like(gues)
    infl = torch.ones(npts * nv, dtype=f64, device=dev)
    status = torch.zeros(npts, dtype=torch.int32, device=dev)
    nsweep = torch.zeros(npts, dtype=torch.int32, device=dev)
    nobs_pt = torch.zeros(npts, dtype=torch.int32, device=dev)
    relax = dict(rtps=dict(relax_alpha_spread=0.95), rtpp=dict(relax_alpha=0.7), none=dict())[args.relax]
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    d64 = lambda v: torch.tensor(v, dtype=f64, device=dev)
    const = dict(group_start=i32([0, 1]), group_member=i32([0]), vmode=i32([1]), hori_loc=d64([hloc]),
                 vert_loc=d64([vloc]), varloc=d64([1.0]), max_nobs=i32([0]), ngrd_i=i32([ngi]), ngrd_j=i32([ngj]),
                 ngrdsch_i=i32([nsi]), ngrdsch_j=i32([nsj]), ngrdext_i=i32([ngi + 2 * nsi]),
                 ngrdext_j=i32([ngj + 2 * nsj]), ac_off=torch.zeros(1, dtype=torch.int64, device=dev))
    ncomm = None
    if world > 1 and args.exchange == "lib":
        import ctypes as C

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        uid = UniqueId()
        if rank == 0:
            assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
        ub = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=dev)
        dist.broadcast(ub, 0)
        C.memmove(C.byref(uid), bytes(ub.cpu().tolist()), 128)
        ncomm = C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        assert rccl.ncclCommInitRank(C.byref(ncomm), world, uid, rank) == 0
    keep = {}
    # warm-start runs up the columns where the vertical neighbour is the closer one (bench.py --warm-runs)
    zl = bw.level_heights(nz, cfg["ztop"])
    zdir = args.warm_runs == "z" or (args.warm_runs == "auto" and nz > 1 and float(np.mean(np.diff(zl))) / vloc < dx / hloc)
    wstride = nij1 if zdir else 0

    def step():
        # ---- set_letkf_obs: sort, exchange, extended subdomain
        n_cell, key = ctx.obs_mesh_sort(mesh, ncell, ctype, ri_l, rj_l, qc)
        send = loc[key.long()].contiguous()
        if world > 1:
            cells = [torch.zeros_like(n_cell) for _ in range(world)]
            dist.all_gather(cells, n_cell.contiguous())
            n_all = torch.stack(cells).contiguous()
            counts = [int(c.sum().item()) for c in cells]
            if args.exchange == "halo":
                bufr = None
            elif ncomm is not None:
                bufr = torch.empty(sum(counts), ncols, dtype=f64, device=dev)
                ctx.obs_allgatherv(ncomm.value, rank, counts, send, bufr)
            else:
                bufr, _ = sharding.allgatherv_rows(send)
        else:
            n_all, bufr, counts = n_cell[None].contiguous(), send, [send.shape[0]]
        ntot = sum(counts)
        ac, src = ctx.obs_halo_plan(lay, n_all, nacx, ntot)
        nt = src.numel()
        tab = torch.empty(nt, ncols, dtype=f64, device=dev)
        if bufr is not None:
            ctx.obs_gather_rows(src, ncols, bufr, ncols, tab, ncols)
        else:
            # halo-only exchange: the cell counts (above) are all a rank needs to know WHICH rows of the virtual gathered
            # table every extended subdomain holds, its own (src) and everybody else's -- so each rank runs the plan once
            # per destination, sends its own rows of it in plan order, and files what it receives in its own plan's order
            offs = np.concatenate([[0], np.cumsum(counts)])
            blocks = []
            for d in range(world):
                if d == rank:
                    src_d = src
                else:
                    lay_d = pkg.HaloLayout()
                    lay_d.nctype, lay_d.nprocs, lay_d.prc_num_x, lay_d.myrank = 1, world, px, d
                    lay_d.ngrd_i, lay_d.ngrd_j, lay_d.ngrdsch_i, lay_d.ngrdsch_j = (gi.ctypes.data, gj.ctypes.data,
                                                                                    si.ctypes.data, sj.ctypes.data)
                    _, src_d = ctx.obs_halo_plan(lay_d, n_all, nacx, ntot)
                own = src_d[(src_d >= int(offs[rank])) & (src_d < int(offs[rank + 1]))].long() - int(offs[rank])
                blocks.append(send[own].contiguous())
            src_rank = torch.bucketize(src.long(), torch.as_tensor(offs[1:], device=dev), right=True)
            rc = torch.bincount(src_rank, minlength=world).tolist()
            got = sharding.exchange_rows(blocks, rc)
            for q in range(world):
                if rc[q]:
                    tab[src_rank == q] = got[q]
            keep["rows_received"] = int(sum(rc) - rc[rank])
        ens = tab[:, :kld].contiguous()
        dep = tab[:, kld].contiguous()
        t = pkg.SearchTables()
        t.nctype, t.ngroup, t.criterion, t.nlon, t.nlat, t.limit_hint = 1, 1, 1, nlon, nlat, 1
        t.dx, t.dy, t.i_org, t.j_org, t.rain_base = dx, dx, 0.5 + pi * nlon, 0.5 + pj * nlat, 8.5e4
        k2 = dict(const, ac_ext=ac, ob_ri=tab[:, kld + 1].contiguous(), ob_rj=tab[:, kld + 2].contiguous(),
                  ob_lev=tab[:, kld + 3].contiguous(), ob_dat=torch.full((max(nt, 1),), 1.0e5, dtype=f64, device=dev),
                  ob_err=torch.full((max(nt, 1),), err, dtype=f64, device=dev))
        for name, v in k2.items():
            setattr(t, name, v.data_ptr())
        # ---- das_letkf: obs_local for the tile's points, then the loop body (--lists pipeline: ONE call of the library, the
        # lists by level slabs or -- where they would not fit --list-gb -- not at all: BASELINE configs[3]'s tile needs that)
        if args.lists == "pipeline":
            ctx.das_columns(k, nv, t, nij1, nz, rig, rjg, prl, prz, ens, kld, dep, infl, gues, anal, 1, npts, npts * nens,
                            list_bytes=int(args.list_gb * 2 ** 30), nobs_out=nobs_pt, status=status, nsweep=nsweep, **relax)
            keep.update(k2=k2, nt=nt, nnz=int(nobs_pt.sum(dtype=torch.int64).item()))
        else:
            off, idx, rd, rl = ctx.obs_search_columns(t, nij1, nz, rig, rjg, prl, prz)
            ctx.das_points(k, nv, off, idx, rd, rl, ens, kld, dep, infl, gues, anal, 1, npts, npts * nens, status=status,
                           nsweep=nsweep, warm_stride=wstride, **relax)
            keep.update(k2=k2, nt=nt, nnz=int(off[-1].item()))
        if "rows_received" not in keep:
            keep["rows_received"] = int(ntot - counts[rank]) if world > 1 else 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(True)
    ctx.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, nlaunch = ctx.timing_read(reset=True)
    csum = float(anal.view(nv, nens, npts)[:, :k].sum().item())          # (slot k, the mean, is not written by the loop)
    stat = torch.tensor([elapsed, float(keep["nnz"]), float(keep["nt"]), float((status != 0).sum().item()),
                         float(nsweep.clamp(min=0).double().sum().item()), kern_ms, float(keep["rows_received"]), csum], dtype=f64,
                        device=dev)
    mx = stat.clone()
    if world > 1:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
    elapsed = float(mx[0].item())
    ntot = npts * world
    n_mean = float(stat[1].item()) / ntot
    return dict(elapsed=elapsed, npts_total=ntot, n_mean=n_mean, halo_rows_mean=float(stat[2].item()) / world,
                bad=int(stat[3].item()), sweeps_mean=float(stat[4].item()) / ntot, kern_ms=float(mx[5].item()),
                nlaunch=nlaunch, k=k, nv=nv, tiles=f"{px}x{py} tiles of {nlon}x{nlat}x{nz}", nobs=nobs,
                rows_received_mean=float(stat[6].item()) / world, anal_checksum=float(stat[7].item()))
