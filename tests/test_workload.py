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


def test_c1_builder_lists_are_the_oracles_obs_local():
    """bench_workload "C1" (500 random conventional observations, ln-p vertical localisation): the
    brute-force lists of the torch builder = orc_obs_local (scale/letkf/letkf_tools.f90:1325-1759) on the tables of
    search_tables(), row for row and weight for weight, on a sample of points -- and the tie flag of the oracle's limited
    selection is silent without a limit."""
    import numpy as np
    import torch
    import _search
    import bench_workload as bw
    from __graft_entry__ import load_package
    pkg = load_package()
    dev = torch.device("cpu")
    w = bw.build("C1", dev)
    assert w["npts"] == 48000 and w["nobs"] == 500 and w["n_max"] == 500 and 350 < w["n_mean"] < 450
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev)
    h, alive = _search.host_struct_from_torch(t_s, keep)
    rng = np.random.default_rng(0)
    sample = np.sort(rng.choice(w["npts"], 300, replace=False))
    P = [p.numpy() for p in pts]
    off, idx, rd, rl, tied = _search.oracle_csr(h, P[0][sample], P[1][sample], P[2][sample], P[3][sample])
    assert not tied.any()
    mapped, ok = bw.remap_lists_to_sorted(w["obs_idx"], order, w["ensval"].shape[0])
    assert ok
    for i, p in enumerate(sample):
        a, b = int(w["obs_off"][p]), int(w["obs_off"][p + 1])
        assert np.array_equal(mapped[a:b].numpy(), idx[off[i]:off[i + 1]])
        assert np.allclose(w["rloc"][a:b].numpy(), rl[off[i]:off[i + 1]], rtol=1e-14, atol=0)
        assert np.allclose(w["rdiag"][a:b].numpy(), rd[off[i]:off[i + 1]], rtol=1e-14, atol=0)


def test_oracle_reports_ties_of_the_limited_selection():
    """orc_obs_local_tied: with MAX_NOBS_PER_GRID on a regular lattice a grid point that sits symmetrically between observations
    has equal keys at the threshold -- which of them the reference keeps is up to its unstable quick-select
    (common/common_sort.f90:341-369), so a checker must know; at a generic position there is no tie."""
    import numpy as np
    import torch
    import _search
    import bench_workload as bw
    from __graft_entry__ import load_package
    pkg = load_package()
    dev = torch.device("cpu")
    w = bw.build("C2-mini", dev, lists=False)
    t_s, keep, order, pts = bw.search_tables(w, pkg, dev, max_nobs=30)
    h, alive = _search.host_struct_from_torch(t_s, keep)
    P = [p.numpy() for p in pts]
    sel = np.arange(0, 400)
    off, idx, rd, rl, tied = _search.oracle_csr(h, P[0][sel], P[1][sel], P[2][sel], P[3][sel])
    assert (np.diff(off) <= 60).all() and (np.diff(off) == 60).any()
    assert tied.any() and not tied.all()
    # nudged off the lattice's symmetry axes: no ties
    off2, _, _, _, tied2 = _search.oracle_csr(h, P[0][sel] + 0.137, P[1][sel] + 0.291, P[2][sel], P[3][sel] + 13.7)
    assert not tied2.any()
