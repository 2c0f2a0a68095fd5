#!/usr/bin/env python3
"""Bandwidth of the HBM-bound kernels either side of the solve at C2 size:
algorithmic bytes / HIP-event time against the 8 TB/s HBM peak.  Not the contract bench (that is bench.py)."""
import json
import sys
import os

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    pkg = load_package()
    pkg.build()
    dev = torch.device("cuda:0")
    ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
    nx, ny, nz, k, nv = 240, 240, 60, 50, 11
    nens = k + 1
    npts = nx * ny * nz
    x = torch.randn(nv * nens * npts, dtype=torch.float64, device=dev)
    sp, sm, sv = 1, npts, npts * nens
    out = {}
    t = timed(lambda: ctx.ens_mean(k, nv, npts, x, sp, sm, sv))
    out["ens_mean"] = dict(s=t, GBps=8 * npts * nv * (k + 1) / t / 1e9)
    t = timed(lambda: ctx.to_perturbations(k, nv, npts, x, sp, sm, sv))
    out["to_perturbations"] = dict(s=t, GBps=8 * npts * nv * (2 * k + 1) / t / 1e9)
    sprd = torch.empty(npts * nv, dtype=torch.float64, device=dev)
    t = timed(lambda: ctx.ens_spread(k, nv, npts, x, sp, sm, sv, sprd))
    out["ens_spread"] = dict(s=t, GBps=8 * npts * nv * (k + 2) / t / 1e9)
    fld = torch.rand(nv * nx * ny * nz, dtype=torch.float64, device=dev) + 0.5
    t = timed(lambda: ctx.member_points(0, nz, nx, ny, nv, 1, 0, 3, fld, x, nx * ny, sp, sm, sv))
    out["member_to_points"] = dict(s=t, GBps=2 * 8 * npts * nv / t / 1e9)
    t = timed(lambda: ctx.member_points(1, nz, nx, ny, nv, 1, 0, 3, fld, x, nx * ny, sp, sm, sv))
    out["points_to_member"] = dict(s=t, GBps=2 * 8 * npts * nv / t / 1e9)
    c = pkg.scale_rm_consts()
    fld[5 * npts:] *= 1e-3
    t = timed(lambda: ctx.state_trans(c, nz, nx, ny, nv, fld, inverse=False), reps=2)
    out["state_trans"] = dict(s=t, GBps=8 * npts * (11 + 5) / t / 1e9)
    for v in out.values():
        v["frac_of_8TBps"] = v["GBps"] / 8000.0
    print(json.dumps({"grid": "240x240x60, k=50, nv=11", "kernels": out}))


if __name__ == "__main__":
    main()
