"""GPU parity of the set_letkf_obs row (f2) through the C ABI: departure + QC, bucket sort, extended-subdomain plan and
row gathers against the oracle's restatement -- all results bit-identical (integers, and the departures, whose member
sum is sequential on both sides).  The ranks of a multi-subdomain world are run one after the other on the one GPU;
the all-gather between the two halves is a concatenation here (RCCL in bench.py / INTEGRATION.md)."""
import numpy as np
import pytest
import torch

from _obsprep import h08_rows, make_world, mesh_struct, layout_struct, oracle_plan, oracle_rank_stage12, oracle_departure_h08, qc_params

pytestmark = pytest.mark.gpu


def gpu_rank_stage12(w, rk):
    from _gpu import ctx, dev, pkg
    c = ctx()
    ens = dev(rk["ensval"])
    qc = dev(rk["qc"])
    n = len(rk["qc"])
    val = torch.zeros(max(n, 1), dtype=torch.float64, device="cuda")
    c.obs_departure(qc_params(pkg.QcParams, w["k"], w["det_run"]), dev(rk["elm"]), dev(rk["dat"]), dev(rk["err"]), ens,
                    w["kld"], val, qc)
    m = mesh_struct(pkg.Mesh, w, rk)
    n_cell, key = c.obs_mesh_sort(m, w["ncell"], dev(rk["ctype"]), dev(rk["ri"]), dev(rk["rj"]), qc)
    torch.cuda.synchronize()
    return dict(ensval=ens, val=val[:n], qc=qc, n_cell=n_cell, key=key)


@pytest.mark.parametrize("k,det_run,nobs", [(10, True, 3000), (50, False, 20000), (100, True, 1000), (3, False, 65)])
def test_departure_qc_and_sort_bit_exact(k, det_run, nobs):
    w = make_world(11, px=2, py=1, k=k, det_run=det_run, nobs=nobs, ngrd=((4, 6), (6, 3), (3, 4)))
    for rk in w["ranks"]:
        o = oracle_rank_stage12(w, rk)
        g = gpu_rank_stage12(w, rk)
        assert np.array_equal(g["qc"].cpu().numpy(), o["qc"])
        assert np.array_equal(g["ensval"].cpu().numpy(), o["ensval"])
        good = o["qc"] == 0
        assert np.array_equal(g["val"].cpu().numpy()[good], o["val"][good])
        assert np.array_equal(g["n_cell"].cpu().numpy(), o["n_cell"])
        assert np.array_equal(g["key"].cpu().numpy(), o["key"])
        assert len(o["key"]) > 0.5 * len(o["qc"])


def test_empty_and_all_rejected():
    from _gpu import ctx, dev, pkg
    w = make_world(12, px=1, py=1, nobs=200)
    rk = w["ranks"][0]
    rk["qc"][:] = 21
    g = gpu_rank_stage12(w, rk)
    assert g["key"].numel() == 0 and int(g["n_cell"].sum()) == 0
    assert np.array_equal(g["ensval"].cpu().numpy(), rk["ensval"])
    m = mesh_struct(pkg.Mesh, w, rk)
    e = torch.zeros(0, dtype=torch.int32, device="cuda")
    d = torch.zeros(0, dtype=torch.float64, device="cuda")
    n_cell, key = ctx().obs_mesh_sort(m, w["ncell"], e, d, d, e)
    assert key.numel() == 0 and int(n_cell.sum()) == 0


@pytest.mark.parametrize("px,py", [(1, 1), (2, 2), (3, 2)])
def test_extended_subdomain_assembly(px, py):
    from _gpu import ctx, dev, pkg
    c = ctx()
    w = make_world(13, px=px, py=py, nobs=6000, k=12)
    og = [oracle_rank_stage12(w, rk) for rk in w["ranks"]]
    gg = [gpu_rank_stage12(w, rk) for rk in w["ranks"]]
    # "ALLGATHERV": sorted send buffers of every rank, rank-major (letkf_obs.f90:993-1025)
    bufr_ens = torch.cat([g["ensval"][g["key"].long()] for g in gg])
    bufr_val = torch.cat([g["val"][g["key"].long()] for g in gg])
    bufr_gidx = torch.cat([dev(rk["gidx"].astype(np.int32))[g["key"].long()] for rk, g in zip(w["ranks"], gg)])
    n_all = torch.stack([g["n_cell"] for g in gg]).contiguous()
    n_all_o = np.stack([o["n_cell"] for o in og])
    cap = bufr_val.numel()
    for me in range(px * py):
        ac_o, src_o, nt_o = oracle_plan(w, me, n_all_o, cap)
        ac_g, src_g = c.obs_halo_plan(layout_struct(pkg.HaloLayout, w, me), n_all, w["nacx"], cap)
        assert np.array_equal(ac_g.cpu().numpy(), ac_o)
        assert np.array_equal(src_g.cpu().numpy(), src_o)
        nt = src_g.numel()
        ens = torch.full((nt, w["kld"]), float("nan"), dtype=torch.float64, device="cuda")
        val = torch.full((nt,), float("nan"), dtype=torch.float64, device="cuda")
        gid = torch.full((nt,), -1, dtype=torch.int32, device="cuda")
        c.obs_gather_rows(src_g, w["kld"], bufr_ens, w["kld"], ens, w["kld"])
        c.obs_gather_rows(src_g, 1, bufr_val, 1, val, 1)
        c.obs_gather_i32(src_g, bufr_gidx, gid)
        torch.cuda.synchronize()
        assert torch.equal(ens, bufr_ens[src_g.long()])
        assert torch.equal(val, bufr_val[src_g.long()])
        assert torch.equal(gid, bufr_gidx[src_g.long()])
    # too small a row map is reported, not overrun
    with pytest.raises(RuntimeError):
        c.obs_halo_plan(layout_struct(pkg.HaloLayout, w, 0), n_all, w["nacx"], 3)


def test_f2_tables_drive_the_search():
    """The tables produced on the device by the f2 entries (ac_ext, obsda_sort order) are what letkf_obs_search_dev
    consumes: local lists from them equal the oracle's obs_local on the oracle-built tables."""
    import ctypes as C
    import _oracle
    from _gpu import ctx, dev, pkg
    from _search import SearchTables
    c = ctx()
    nlon = nlat = 12
    w = make_world(14, px=2, py=2, nlon=nlon, nlat=nlat, nobs=8000, k=8, det_run=False)
    gg = [gpu_rank_stage12(w, rk) for rk in w["ranks"]]
    n_all = torch.stack([g["n_cell"] for g in gg]).contiguous()
    glob = {f: torch.cat([dev(rk[f])[g["key"].long()] for rk, g in zip(w["ranks"], gg)]) for f in ("ri", "rj")}
    me = 3
    cap = glob["ri"].numel()
    ac_g, src_g = c.obs_halo_plan(layout_struct(pkg.HaloLayout, w, me), n_all, w["nacx"], cap)
    nt = src_g.numel()
    rk = w["ranks"][me]
    # metadata of obsda_sort rows (obs(set)%ri(idx) etc.): gathered with the same row map
    ob = {}
    for f in ("ri", "rj"):
        ob[f] = torch.empty(nt, dtype=torch.float64, device="cuda")
        c.obs_gather_rows(src_g, 1, glob[f], 1, ob[f], 1)
    rng = np.random.default_rng(5)
    ob_lev = dev(rng.uniform(0.0, 12000.0, nt))
    ob_dat = dev(np.full(nt, 1.0e5))
    ob_err = dev(rng.choice([1.0, 2.0], nt))
    nc = w["nctype"]
    gi, gj, si, sj = w["ngrd_i"], w["ngrd_j"], w["ngrdsch_i"], w["ngrdsch_j"]
    ac_off = np.concatenate([[0], np.cumsum((gi + 2 * si + 1).astype(np.int64) * (gj + 2 * sj))])[:nc]
    dx = 1000.0
    arrs = dict(group_start=np.arange(nc + 1, dtype=np.int32), group_member=np.arange(nc, dtype=np.int32),
                vmode=np.ones(nc, np.int32), hori_loc=(si * (dx * nlon / gi) / 3.651483717 * 0.999),
                vert_loc=np.full(nc, 3000.0), varloc=np.ones(nc), max_nobs=np.zeros(nc, np.int32), ngrd_i=gi,
                ngrd_j=gj, ngrdsch_i=si, ngrdsch_j=sj, ngrdext_i=(gi + 2 * si).astype(np.int32),
                ngrdext_j=(gj + 2 * sj).astype(np.int32), ac_off=ac_off.astype(np.int64))
    scal = dict(nctype=nc, ngroup=nc, criterion=1, nlon=nlon, nlat=nlat, dx=dx, dy=dx,
                i_org=w["ihalo"] + 0.5 + rk["pi"] * nlon, j_org=w["ihalo"] + 0.5 + rk["pj"] * nlat, rain_base=8.5e4)
    td, th = pkg.SearchTables(), SearchTables()
    keep = []
    for t, to_dev in ((td, True), (th, False)):
        for k_, v in scal.items():
            setattr(t, k_, v)
        for k_, v in arrs.items():
            a = dev(np.ascontiguousarray(v)) if to_dev else np.ascontiguousarray(v)
            keep.append(a)
            setattr(t, k_, a.data_ptr() if to_dev else a.ctypes.data)
    dev_fields = dict(ac_ext=ac_g, ob_ri=ob["ri"], ob_rj=ob["rj"], ob_lev=ob_lev, ob_dat=ob_dat, ob_err=ob_err)
    for k_, v in dev_fields.items():
        setattr(td, k_, v.data_ptr())
        a = np.ascontiguousarray(v.cpu().numpy())
        keep.append(a)
        setattr(th, k_, a.ctypes.data)
    npts = 60
    pri = scal["i_org"] + rng.uniform(0.5, nlon - 0.5, npts)
    prj = scal["j_org"] + rng.uniform(0.5, nlat - 0.5, npts)
    prz = rng.uniform(0.0, 12000.0, npts)
    prl = np.full(npts, 1.0e5)
    off, idx, rd, rl = c.obs_search(td, dev(pri), dev(prj), dev(prl), dev(prz))
    torch.cuda.synchronize()
    off, idx, rd, rl = off.cpu().numpy(), idx.cpu().numpy(), rd.cpu().numpy(), rl.cpu().numpy()
    lib = _oracle.oracle()
    lib.orc_obs_local.restype = C.c_int
    capl = 20000
    oi, ord_, orl = np.zeros(capl, np.int32), np.zeros(capl), np.zeros(capl)
    total = 0
    for p in range(npts):
        n = lib.orc_obs_local(C.byref(th), C.c_double(pri[p]), C.c_double(prj[p]), C.c_double(prl[p]),
                              C.c_double(prz[p]), C.c_int(capl), oi.ctypes.data_as(C.POINTER(C.c_int32)),
                              ord_.ctypes.data_as(C.POINTER(C.c_double)), orl.ctypes.data_as(C.POINTER(C.c_double)),
                              None)
        assert n == off[p + 1] - off[p]
        assert np.array_equal(idx[off[p]:off[p + 1]], oi[:n])
        assert np.allclose(rl[off[p]:off[p + 1]], orl[:n], rtol=1e-14, atol=0)
        total += n
    assert total > 500


@pytest.mark.parametrize("k,det_run,nobs,val2", [(10, True, 4000, True), (50, False, 3000, True), (20, True, 700, False)])
def test_departure_h08_build_bit_exact(k, det_run, nobs, val2):
    """letkf_qc_params.h08 = 1 = the reference's -DH08 build (letkf_obs.f90:432-469 level / undef rejection and the cloudy-member
    count with the sign restored, :480-487 CA into val2, :520-541 sky-dependent gross-error bound and H08_BT_MIN) against the
    oracle's restatement, bit for bit; h08 = 0 on the same rows = ordinary rows (the default build)."""
    from _gpu import ctx, dev, pkg
    r = h08_rows(77 + k, k, det_run, nobs)
    for h08 in (1, 0):
        over = dict(h08=h08, h08_min_cld_member=2, h08_limit_lev=20000.0, gross_error_h08=4.0, h08_bt_min=180.0)
        ens_o, val_o, qc_o, v2_o = oracle_departure_h08(r, k, det_run, val2, **over)
        ens_g, qc_g, v2_g, lev_g = dev(r["ens"]), dev(r["qc"]), dev(r["val2"]), dev(r["lev"])
        val_g = torch.zeros(nobs, dtype=torch.float64, device="cuda")
        pg = qc_params(pkg.QcParams, k, det_run, **over)
        pg.h08_lev = lev_g.data_ptr()
        pg.h08_val2 = v2_g.data_ptr() if val2 else None
        ctx().obs_departure(pg, dev(r["elm"]), dev(r["dat"]), dev(r["err"]), ens_g, r["kld"], val_g, qc_g)
        torch.cuda.synchronize()
        assert np.array_equal(qc_g.cpu().numpy(), qc_o)
        assert np.array_equal(ens_g.cpu().numpy(), ens_o)
        good = qc_o == 0
        assert np.array_equal(val_g.cpu().numpy()[good], val_o[good])
        assert np.array_equal(v2_g.cpu().numpy(), v2_o)
        hq = qc_o[r["h08"] & (r["qc"] == 0)]
        if h08:
            assert (hq == 50).sum() > 10 and (hq == 5).sum() > 10 and (hq == 0).sum() > 10     # every branch is taken
            assert not np.array_equal(v2_o, r["val2"]) or not val2
        else:
            assert (hq == 50).sum() == 0 and np.array_equal(v2_o, r["val2"])
