"""bench_workload's torch-built local-observation lists against a brute-force search that applies the oracle's
obs_local_cal (scale/letkf/letkf_tools.f90:1793-1906) to every observation.  CPU only."""
import ctypes as C

import numpy as np
import torch

import _oracle
import bench_workload as bw


def test_local_lists_match_obs_local_cal():
    w = bw.build("C2-mini", torch.device("cpu"))
    cfg = w["cfg"]
    nx, ny, nz = cfg["nx"], cfg["ny"], cfg["nz"]
    sp = cfg["spacing"]
    ox = np.arange(0.5 * sp, nx * cfg["dx"], sp)
    oy = np.arange(0.5 * sp, ny * cfg["dx"], sp)
    oz = np.arange(0.5 * sp, cfg["ztop"], sp)
    zlev = bw.level_heights(nz, cfg["ztop"])
    lib = _oracle.oracle()
    off = w["obs_off"].numpy()
    idx = w["obs_idx"].numpy()
    rloc = w["rloc"].numpy()
    rdiag = w["rdiag"].numpy()
    rng = np.random.default_rng(0)
    nd, nr = C.c_double(), C.c_double()
    for pt in rng.choice(w["npts"], size=12, replace=False):
        lev, ij = divmod(int(pt), nx * ny)
        j, i = divmod(ij, nx)
        # work in grid-index space with DX: ri = x/dx (any common origin cancels in letkf_tools.f90:1876-1877)
        ri, rj, rz = (i + 0.5), (j + 0.5), zlev[lev]
        exp = []
        for iob in range(w["nobs"]):
            iz, rem = divmod(iob, len(oy) * len(ox))
            iy, ix = divmod(rem, len(ox))
            r = lib.orc_obs_local_cal(C.c_double(ri), C.c_double(rj), C.c_double(1.0e5), C.c_double(rz),
                                      C.c_double(1.0), C.c_int(1), C.c_double(cfg["hloc"]), C.c_double(cfg["vloc"]),
                                      C.c_double(0.0), C.c_double(ox[ix] / cfg["dx"]), C.c_double(oy[iy] / cfg["dx"]),
                                      C.c_double(oz[iz]), C.c_double(0.0), C.c_double(cfg["err"]),
                                      C.c_double(cfg["dx"]), C.c_double(cfg["dx"]), C.byref(nd), C.byref(nr))
            if r != 0.0:
                exp.append((iob, r, nr.value))
        got = idx[off[pt]:off[pt + 1]]
        assert got.tolist() == [e[0] for e in exp], pt
        assert np.allclose(rloc[off[pt]:off[pt + 1]], [e[1] for e in exp], rtol=1e-12, atol=0)
        assert np.allclose(rdiag[off[pt]:off[pt + 1]], [e[2] for e in exp], rtol=1e-12, atol=0)


def test_remap_lists_to_sorted_table():
    """bench.py --lists pipeline / fused: the CPU checker's lists follow the table through its mesh sort (a subset of the rows on
    the disc workloads); an entry whose row the sorted table lost is reported, never used as an index."""
    import torch
    import bench_workload as bw
    order = torch.tensor([5, 2, 7, 0], dtype=torch.int64)          # sorted row i = original row order[i]; rows 1, 3, 4, 6 lost
    idx = torch.tensor([0, 7, 2, 5, 5], dtype=torch.int32)
    mapped, ok = bw.remap_lists_to_sorted(idx, order, 8)
    assert ok and mapped.tolist() == [3, 2, 1, 0, 0] and mapped.dtype == torch.int32
    mapped, ok = bw.remap_lists_to_sorted(torch.tensor([0, 3], dtype=torch.int32), order, 8)
    assert not ok and mapped.tolist() == [3, 0]
    # a second combined type: the table twice, the second copy's rows shifted by the table length
    order2 = torch.cat([order, order + 8])
    mapped, ok = bw.remap_lists_to_sorted(torch.tensor([5, 13], dtype=torch.int32), order2, 16)
    assert ok and mapped.tolist() == [0, 4]
