"""Run by tests/test_gpu_checked.py in a process of its own with LETKF_AMD_LIB = the CHECKED twin of the library (make CHECKED=1:
every index the column-survivor mode of the loop-body kernel derives from device data is tested against the bound the host sized
its buffers by, a violation comes back as an error of the entry instead of a memory fault).  Drives the list-free route of
letkf_das_columns_dev over the shapes that stress those bounds: many small batches of columns (ragged last one), a dense disc with
empty columns, beta zeros, odd level counts and short runs.  Prints one line per case; exit status 0 = no bound was touched."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import bench_workload as bw
    from __graft_entry__ import load_package
    pkg = load_package()
    assert "checked" in pkg.LIB_PATH, pkg.LIB_PATH
    dev = torch.device("cuda:0")
    cx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
    cx.set_option(cx.OPT_COLUMN_SURVIVORS, 1)
    cases = [("C2-mini", 0.5, None, 0), ("C2-mini", 0.05, 7, 5), ("C2-mini-disc", 1.0, 11, 3), ("C2-mini-k20", 64.0, None, 0),
             ("C4-slab-disc", 40.0, None, 0), ("C4-slab", 24.0, 5, 2)]
    for name, list_mb, nlev_cut, run in cases:
        w = bw.build(name, dev, det_run=True, lists=False)
        k, nv, nens, npts_all = w["k"], w["nv"], w["nens"], w["npts"]
        nij1 = w["cfg"]["nx"] * w["cfg"]["ny"]
        nlev = nlev_cut or w["cfg"]["nz"]
        npts = nij1 * nlev
        cx.ens_mean(k, nv, npts_all, w["gues"], 1, npts_all, npts_all * nens)
        cx.to_perturbations(k, nv, npts_all, w["gues"], 1, npts_all, npts_all * nens)
        t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
        ens, dep = w["ensval"][order].contiguous(), w["dep"][order].contiguous()
        rig, rjg = pts[0][:nij1].contiguous(), pts[1][:nij1].contiguous()
        g = torch.Generator(device=dev)
        g.manual_seed(17)
        beta = torch.rand(npts, generator=g, device=dev, dtype=torch.float64)
        beta[torch.rand(npts, generator=g, device=dev) < 0.2] = 0.0
        anal = torch.full_like(w["gues"], float("nan"))
        infl = torch.ones(npts_all * nv, dtype=torch.float64, device=dev)
        status = torch.full((npts,), -1, dtype=torch.int32, device=dev)
        nobs = torch.full((npts,), -1, dtype=torch.int32, device=dev)
        cx.das_columns(k, nv, t_s, nij1, nlev, rig, rjg, pts[2][:npts].contiguous(), pts[3][:npts].contiguous(), ens, w["kld"], dep,
                       infl, w["gues"], anal, 1, npts_all, npts_all * nens, list_bytes=int(list_mb * 2 ** 20), nobs_out=nobs, beta=beta,
                       det_run=True, infl_adaptive=True, relax_alpha_spread=0.95, status=status, warm_run=run, infl_sv=npts_all)
        torch.cuda.synchronize()
        assert "FUSED" in cx.last_path(), cx.last_path()
        assert int(status.abs().max()) == 0
        print(f"checked ok: {name} list_mb={list_mb} nlev={nlev} run={run}: n up to {int(nobs.max())}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
