#!/usr/bin/env python3
"""Stage-by-stage wall time of one analysis on one MI355X at C2 size through the C ABI, in the order the reference's
PROGRAM letkf runs them (scale/letkf/letkf.f90): set_letkf_obs (departures + QC, bucket sort, extended-subdomain plan,
obsda_sort gathers) -> das_letkf (perturbation pass, obs_local for every point, the loop body, additive inflation) ->
ensmean_grd / departure statistics.  Synthetic data (bench_workload.C2); everything device-resident.  Not the contract
bench (bench.py): this one answers "where does an analysis cycle spend its time"."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import bench_workload as bw                # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


def main():
    pkg = load_package()
    pkg.build()
    dev = torch.device("cuda:0")
    ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    w = bw.build(name, dev)
    cfg = w["cfg"]
    k, nv, npts, kld = w["k"], w["nv"], w["npts"], w["kld"]
    nij, nlev = cfg["nx"] * cfg["ny"], cfg["nz"]
    sp, sm, sv = w["sp"], w["sm"], w["sv"]
    stages = {}

    # ---- set_letkf_obs: the workload's table plays H(x) of the members; one rank, one radar ctype
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    nobs = w["nobs"]
    hx = (w["ensval"] + 20.0).contiguous()
    elm = torch.full((nobs,), 4002, dtype=torch.int32, device=dev)        # radial velocity
    dat = torch.full((nobs,), 20.0, dtype=torch.float64, device=dev) + w["dep"]
    err = torch.full((nobs,), cfg["err"], dtype=torch.float64, device=dev)
    qp = pkg.QcParams(member=k, det_run=0, use_radar_ref=1, use_radar_vr=1, min_radar_ref_member=1,
                      min_radar_ref_member_obsref=1, radar_ref_thres_dbz=15.0, gross_error=1e9, gross_error_rain=1e9,
                      gross_error_radar_ref=1e9, gross_error_radar_vr=1e9, gross_error_radar_prh=1e9,
                      gross_error_tcx=1e9, gross_error_tcy=1e9, gross_error_tcp=1e9)
    val = torch.zeros(nobs, dtype=torch.float64, device=dev)
    qc = torch.zeros(nobs, dtype=torch.int32, device=dev)

    def departure():
        e = hx.clone()
        q = qc.clone()
        ctx.obs_departure(qp, elm, dat, err, e, kld, val, q)
        return e, q
    stages["obs_departure_qc"], (ens_d, qc_d) = timed(departure)
    ngi, ngj = int(keep["ngrd_i"][0]), int(keep["ngrd_j"][0])
    nsi, nsj = int(keep["ngrdsch_i"][0]), int(keep["ngrdsch_j"][0])
    gi, gj = np.array([ngi], np.int32), np.array([ngj], np.int32)
    si, sj = np.array([nsi], np.int32), np.array([nsj], np.int32)
    mesh = pkg.Mesh(nctype=1, nlon=cfg["nx"], nlat=cfg["ny"], ihalo=0, jhalo=0, rank_i=0, rank_j=0,
                    ngrd_i=gi.ctypes.data, ngrd_j=gj.ctypes.data)
    # lattice coordinates of the table rows (the reference's ri, rj carry a 0.5 offset: grid point i sits at i)
    ob_ri = torch.empty(nobs, dtype=torch.float64, device=dev)
    ob_rj = torch.empty(nobs, dtype=torch.float64, device=dev)
    ob_ri[order] = keep["ob_ri"] + 0.5
    ob_rj[order] = keep["ob_rj"] + 0.5
    ctype = torch.zeros(nobs, dtype=torch.int32, device=dev)
    stages["obs_mesh_sort"], (n_cell, key) = timed(lambda: ctx.obs_mesh_sort(mesh, ngi * ngj, ctype, ob_ri, ob_rj, qc_d))
    lay = pkg.HaloLayout(nctype=1, nprocs=1, prc_num_x=1, myrank=0, ngrd_i=gi.ctypes.data, ngrd_j=gj.ctypes.data,
                         ngrdsch_i=si.ctypes.data, ngrdsch_j=sj.ctypes.data)
    nacx = (ngi + 2 * nsi + 1) * (ngj + 2 * nsj)
    n_all = n_cell.view(1, -1).contiguous()
    stages["obs_halo_plan"], (ac_ext, src_row) = timed(lambda: ctx.obs_halo_plan(lay, n_all, nacx, nobs))
    bufr = ens_d[key.long()].contiguous()
    sort_ens = torch.empty(src_row.numel(), kld, dtype=torch.float64, device=dev)
    stages["obsda_sort_gather"], _ = timed(lambda: ctx.obs_gather_rows(src_row, kld, bufr, kld, sort_ens, kld))
    assert torch.equal(ac_ext, keep["ac_ext"]), "device-built ac_ext differs from the workload builder's"

    # ---- das_letkf
    gues = w["gues"]
    stages["ensmean_grd(gues)"], _ = timed(lambda: ctx.ens_mean(k, nv, npts, gues, sp, sm, sv))
    g2 = gues.clone()
    stages["perturbation_pass"], _ = timed(lambda: ctx.to_perturbations(k, nv, npts, g2, sp, sm, sv), reps=1)
    ctx.to_perturbations(k, nv, npts, gues, sp, sm, sv)
    del g2
    ens_sorted = w["ensval"][order].contiguous()
    dep_sorted = w["dep"][order].contiguous()
    rig, rjg = pts[0][:nij].contiguous(), pts[1][:nij].contiguous()
    stages["obs_local_all_points"], lists = timed(lambda: ctx.obs_search_columns(t_s, nij, nlev, rig, rjg, pts[2], pts[3]))
    off, idx, rd, rl = lists
    anal = torch.empty_like(gues)
    infl = torch.ones(npts * nv, dtype=torch.float64, device=dev)
    st = torch.zeros(npts, dtype=torch.int32, device=dev)
    stages["loop_body(letkf_core+transform)"], _ = timed(
        lambda: ctx.das_points(k, nv, off, idx, rd, rl, ens_sorted, kld, dep_sorted, infl, gues, anal, sp, sm, sv,
                               status=st, relax_alpha_spread=0.95, warm_stride=nij))   # warm-start runs up the columns, as bench.py
    assert int(st.abs().max()) == 0
    del off, idx, rd, rl, lists
    add = torch.randn_like(gues)
    ctx.ens_mean(k, nv, npts, add, sp, sm, sv)
    ctx.to_perturbations(k, nv, npts, add, sp, sm, sv)
    stages["additive_inflation"], _ = timed(lambda: ctx.additive_inflation(k, nv, npts, nij, anal, add, sp, sm, sv, 0.1))
    del add
    stages["ensmean_grd(anal)"], _ = timed(lambda: ctx.ens_mean(k, nv, npts, anal, sp, sm, sv))
    ids = np.array([2819, 2820, 3073, 3074, 3330, 3331, 14593, 19999, 4001, 4004, 4002, 4003, 8800, 99991, 99992, 99993],
                   dtype=np.int32)
    stages["monit_dep"], _ = timed(lambda: ctx.monit_dep(ids, elm, val, qc_d))
    total = sum(stages.values())
    print(json.dumps({"workload": f"{name}: {cfg['nx']}x{cfg['ny']}x{cfg['nz']}, k={k}, nv={nv}, {nobs} obs rows, "
                                  f"mean {w['n_mean']:.0f} local obs/point", "stage_ms": stages,
                      "sum_ms": total, "points": npts, "points_per_s_whole_cycle": npts / (total * 1e-3)}))


if __name__ == "__main__":
    main()
