"""One analysis over a domain cut into px x py subdomains ("virtual ranks" run one after the other on the one GPU):
the set_letkf_obs pipeline per rank (departure + QC -> mesh sort -> [all-gather = concatenation] -> extended-subdomain
plan -> row gathers; scale/letkf/letkf_obs.f90:361-1138), the search tables it leaves behind, obs_local + the loop
body for the rank's own grid points.  Used to require that a tiled analysis equals the single-domain one."""
import numpy as np
import torch

from _obsprep import layout_struct, make_world, mesh_struct, qc_params

DZF = float(np.float32(3.651483717))


def rank_pipeline(w, lev_glob, dx):
    """Runs the f2 entries for every rank of world `w`.  Returns per rank: dict(tables, keep, ensval, dep, gid)
    where gid[r] = global observation number of obsda_sort row r."""
    from _gpu import ctx, dev, pkg
    c = ctx()
    stage = []
    for rk in w["ranks"]:
        ens = dev(rk["ensval"])
        qc = dev(rk["qc"])
        n = len(rk["qc"])
        val = torch.zeros(max(n, 1), dtype=torch.float64, device="cuda")
        c.obs_departure(qc_params(pkg.QcParams, w["k"], w["det_run"]), dev(rk["elm"]), dev(rk["dat"]), dev(rk["err"]),
                        ens, w["kld"], val, qc)
        n_cell, key = c.obs_mesh_sort(mesh_struct(pkg.Mesh, w, rk), w["ncell"], dev(rk["ctype"]), dev(rk["ri"]),
                                      dev(rk["rj"]), qc)
        stage.append(dict(ensval=ens, val=val[:n], key=key, n_cell=n_cell))
    # the path's one exchange (MPI_ALLGATHERV of the sorted buffers + the cell counts, letkf_obs.f90:826-831,1036-1046)
    kl = [s["key"].long() for s in stage]
    bufr = dict(ens=torch.cat([s["ensval"][q] for s, q in zip(stage, kl)]),
                val=torch.cat([s["val"][q] for s, q in zip(stage, kl)]),
                gid=torch.cat([dev(rk["gidx"].astype(np.int32))[q] for rk, q in zip(w["ranks"], kl)]))
    for f in ("ri", "rj", "err"):
        bufr[f] = torch.cat([dev(rk[f])[q] for rk, q in zip(w["ranks"], kl)])
    n_all = torch.stack([s["n_cell"] for s in stage]).contiguous()
    cap = bufr["val"].numel()
    nc = w["nctype"]
    gi, gj, si, sj = w["ngrd_i"], w["ngrd_j"], w["ngrdsch_i"], w["ngrdsch_j"]
    ac_off = np.concatenate([[0], np.cumsum((gi + 2 * si + 1).astype(np.int64) * (gj + 2 * sj))])[:nc]
    lev_d = dev(lev_glob)
    out = []
    for me, rk in enumerate(w["ranks"]):
        ac, src = c.obs_halo_plan(layout_struct(pkg.HaloLayout, w, me), n_all, w["nacx"], cap)
        nt = src.numel()
        g = {}
        for f, ncol in (("ens", w["kld"]), ("val", 1), ("ri", 1), ("rj", 1), ("err", 1)):
            g[f] = torch.empty((nt, ncol) if ncol > 1 else (nt,), dtype=torch.float64, device="cuda")
            c.obs_gather_rows(src, ncol, bufr[f], ncol, g[f], ncol)
        gid = torch.empty(nt, dtype=torch.int32, device="cuda")
        c.obs_gather_i32(src, bufr["gid"], gid)
        t = pkg.SearchTables()
        scal = dict(nctype=nc, ngroup=nc, criterion=1, nlon=w["nlon"], nlat=w["nlat"], dx=dx, dy=dx,
                    i_org=w["ihalo"] + 0.5 + rk["pi"] * w["nlon"], j_org=w["ihalo"] + 0.5 + rk["pj"] * w["nlat"],
                    rain_base=8.5e4)
        for k_, v in scal.items():
            setattr(t, k_, v)
        arrs = dict(group_start=np.arange(nc + 1, dtype=np.int32), group_member=np.arange(nc, dtype=np.int32),
                    vmode=np.ones(nc, np.int32), hori_loc=w["hori_loc"], vert_loc=np.full(nc, 3000.0),
                    varloc=np.ones(nc), max_nobs=np.zeros(nc, np.int32), ngrd_i=gi, ngrd_j=gj, ngrdsch_i=si,
                    ngrdsch_j=sj, ngrdext_i=(gi + 2 * si).astype(np.int32), ngrdext_j=(gj + 2 * sj).astype(np.int32),
                    ac_off=ac_off.astype(np.int64))
        keep = []
        for k_, v in arrs.items():
            a = dev(np.ascontiguousarray(v))
            keep.append(a)
            setattr(t, k_, a.data_ptr())
        devf = dict(ac_ext=ac, ob_ri=g["ri"], ob_rj=g["rj"], ob_lev=lev_d[gid.long()].contiguous(),
                    ob_dat=torch.full((max(nt, 1),), 1.0e5, dtype=torch.float64, device="cuda"), ob_err=g["err"])
        for k_, v in devf.items():
            keep.append(v)
            setattr(t, k_, v.data_ptr())
        out.append(dict(tables=t, keep=keep, ensval=g["ens"], dep=g["val"], gid=gid, nrows=nt))
    torch.cuda.synchronize()
    return out


def tiled_analysis(seed, px, py, nlon_g, nlat_g, nlev, k, nobs, x_glob, zlev, ngrd_cell=(4, 2, 4), nsch=(2, 3, 1),
                   dx=1000.0, relax=None, det_run=True, fix_ij_obsgrd=False):
    """Analysis of the whole (nlon_g x nlat_g x nlev) grid through px x py virtual ranks.  x_glob: numpy
    [nv, nens, nlev, nlat_g, nlon_g] first guess (members, mean slot, det slot).  Returns dict(anal [same shape],
    lists {global point -> (global obs ids, rdiag, rloc)}, nobsl)."""
    from _gpu import ctx, dev
    c = ctx()
    nlon, nlat = nlon_g // px, nlat_g // py
    assert nlon * px == nlon_g and nlat * py == nlat_g
    assert all(nlon % s == 0 and nlat % s == 0 for s in ngrd_cell), "same mesh cells in every decomposition"
    ngrd = tuple((nlon // s, nlat // s) for s in ngrd_cell)
    w = make_world(seed, px=px, py=py, nlon=nlon, nlat=nlat, k=k, det_run=det_run, nobs=nobs, ngrd=ngrd,
                   ngrdsch=tuple((s, s) for s in nsch))
    w["fix_ij_obsgrd"] = int(fix_ij_obsgrd)
    # search radius = what the halo of nsch mesh cells covers (letkf_obs.f90:674-677)
    w["hori_loc"] = np.array([s * cs * dx / DZF * 0.999 for s, cs in zip(nsch, ngrd_cell)])
    lev_glob = np.random.default_rng(seed + 1).uniform(0.0, 12000.0, nobs)
    ranks = rank_pipeline(w, lev_glob, dx)
    nv, nens = x_glob.shape[0], x_glob.shape[1]
    anal = np.full_like(x_glob, np.nan)
    lists = {}
    relax = relax or dict(relax_alpha_spread=0.95)
    for me, (rk, r) in enumerate(zip(w["ranks"], ranks)):
        i0, j0 = rk["pi"] * nlon, rk["pj"] * nlat
        # the rank's points p = ij + nij1*lev, ij = i + nlon*j (gues3d(nij1, nlev, ...)); rig1 = i + IHALO (1-based i)
        ii, jj = np.meshgrid(np.arange(nlon), np.arange(nlat))
        rig = (i0 + ii.ravel() + 1 + w["ihalo"]).astype(np.float64)
        rjg = (j0 + jj.ravel() + 1 + w["ihalo"]).astype(np.float64)
        nij1 = rig.size
        npts = nij1 * nlev
        pri, prj = np.tile(rig, nlev), np.tile(rjg, nlev)
        prz = np.repeat(zlev, nij1)
        prl = np.full(npts, 1.0e5)
        off, idx, rd, rl = c.obs_search(r["tables"], dev(pri), dev(prj), dev(prl), dev(prz))
        xs = x_glob[:, :, :, j0:j0 + nlat, i0:i0 + nlon].reshape(nv, nens, npts)
        gues = dev(np.ascontiguousarray(xs).reshape(-1))
        an = torch.full((gues.numel(),), float("nan"), dtype=torch.float64, device="cuda")
        infl = torch.ones(npts * nv, dtype=torch.float64, device="cuda")
        status = torch.full((npts,), -1, dtype=torch.int32, device="cuda")
        c.das_points(k, nv, off, idx, rd, rl, r["ensval"], w["kld"], r["dep"], infl, gues, an, 1, npts, npts * nens,
                     det_run=det_run, status=status, **relax)
        torch.cuda.synchronize()
        assert int(status.abs().max()) == 0
        anal[:, :, :, j0:j0 + nlat, i0:i0 + nlon] = an.cpu().numpy().reshape(nv, nens, nlev, nlat, nlon)
        off_h, idx_h = off.cpu().numpy(), idx.cpu().numpy()
        gid_h, rd_h, rl_h = r["gid"].cpu().numpy(), rd.cpu().numpy(), rl.cpu().numpy()
        for p in range(npts):
            lev, ij = divmod(p, nij1)
            j, i = divmod(ij, nlon)
            s = slice(off_h[p], off_h[p + 1])
            lists[(lev, j0 + j, i0 + i)] = (gid_h[idx_h[s]], rd_h[s], rl_h[s])
    return dict(anal=anal, lists=lists, world=w, nrows=[r["nrows"] for r in ranks])
