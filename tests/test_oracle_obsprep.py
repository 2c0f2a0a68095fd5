"""CPU checks of the oracle's set_letkf_obs restatement (row f2: scale/letkf/letkf_obs.f90:361-561 departures + QC,
:762-822 bucket sort, :922-976 / :1036-1100 extended-subdomain plan) against independent brute-force formulations."""
import numpy as np
import pytest

from _obsprep import ID_RADAR_REF, ID_RADAR_VR, UNDEF, h08_rows, make_world, oracle_departure_h08, oracle_plan, oracle_rank_stage12


def test_departure_and_qc_rules():
    w = make_world(1, px=1, py=1, nobs=4000)
    rk = w["ranks"][0]
    o = oracle_rank_stage12(w, rk)
    k, kld = w["k"], w["kld"]
    e0, qc0 = rk["ensval"], rk["qc"]
    seen = set()
    for n in range(len(qc0)):
        if qc0[n] > 0:                                   # untouched rows
            assert o["qc"][n] == qc0[n] and np.array_equal(o["ensval"][n], e0[n])
            continue
        el, d = rk["elm"][n], rk["dat"][n]
        if el == ID_RADAR_REF:
            if d == UNDEF:
                assert o["qc"][n] == 50
                seen.add("undef")
                continue
            mem = int((e0[n, :k] > 15.0 + 1e-6).sum())
            need = 2 if d > 15.0 + 1e-6 else 3
            if mem < need:
                assert o["qc"][n] == 12 and np.array_equal(o["ensval"][n], e0[n])
                seen.add("refmem")
                continue
        mean = e0[n, 0]
        for i in range(1, k):
            mean = mean + e0[n, i]
        mean = mean / k
        assert np.array_equal(o["ensval"][n, :k], e0[n, :k] - mean)
        assert o["val"][n] == d - mean
        assert o["ensval"][n, k] == d - e0[n, k]         # DET_RUN column
        ge = {ID_RADAR_REF: 3.0, ID_RADAR_VR: 2.5}.get(el, 5.0)
        gross = abs(d - mean) > ge * rk["err"][n]
        assert o["qc"][n] == (5 if gross else 0)
        seen.add("gross" if gross else "good")
    assert seen == {"undef", "refmem", "gross", "good"}


@pytest.mark.parametrize("ngrd", [((4, 4), (6, 6), (3, 3)), ((4, 6), (6, 3), (3, 4))])
def test_bucket_sort_is_the_stable_counting_sort(ngrd):
    w = make_world(2, px=2, py=1, ngrd=ngrd)
    for rk in w["ranks"]:
        o = oracle_rank_stage12(w, rk)
        good = np.nonzero(o["qc"] == 0)[0]
        assert sorted(o["key"].tolist()) == good.tolist()
        # cell of every row, written independently (note the reference's ngrd_i in the j formula, :1200)
        gi, gj = w["ngrd_i"][rk["ctype"]], w["ngrd_j"][rk["ctype"]]
        ril = rk["ri"] - rk["pi"] * w["nlon"] - w["ihalo"] - 0.5
        rjl = rk["rj"] - rk["pj"] * w["nlat"] - w["ihalo"] - 0.5
        ci = np.clip(np.ceil(ril * gi / w["nlon"]).astype(int), 1, gi)
        cj = np.clip(np.ceil(rjl * gi / w["nlat"]).astype(int), 1, gj)
        coff = np.concatenate([[0], np.cumsum(w["ngrd_i"].astype(int) * w["ngrd_j"])])
        cell = coff[rk["ctype"]] + (cj - 1) * gi + (ci - 1)
        order = good[np.lexsort((good, cell[good]))]
        assert np.array_equal(o["key"], order)
        assert np.array_equal(o["n_cell"], np.bincount(cell[good], minlength=w["ncell"]))


@pytest.mark.parametrize("px,py", [(1, 1), (2, 2), (3, 2)])
def test_extended_subdomain_plan_matches_global_brute_force(px, py):
    w = make_world(3, px=px, py=py, nobs=5000)
    st = [oracle_rank_stage12(w, rk) for rk in w["ranks"]]
    n_all = np.stack([s["n_cell"] for s in st])
    # the ALLGATHERV receive buffer: every rank's sorted rows, rank-major; remember the global obs number of each row
    bufr_gidx = np.concatenate([rk["gidx"][s["key"]] for rk, s in zip(w["ranks"], st)])
    g = w["glob"]
    for me, rk in enumerate(w["ranks"]):
        ac_ext, src_row, nt = oracle_plan(w, me, n_all, cap=len(bufr_gidx))
        assert nt == len(src_row) and nt >= len(st[me]["key"])
        got = bufr_gidx[src_row]
        # brute force: for every ctype, every extended cell, the accepted global obs that fall into it, in
        # (owner rank, local row) = (owner rank, global number) order
        off = 0
        pos = 0
        for ic in range(w["nctype"]):
            gi, gj, si, sj = (int(w[f][ic]) for f in ("ngrd_i", "ngrd_j", "ngrdsch_i", "ngrdsch_j"))
            ei, ej = gi + 2 * si, gj + 2 * sj
            ac = ac_ext[off:off + (ei + 1) * ej].reshape(ej, ei + 1)
            off += (ei + 1) * ej
            sel = np.nonzero(g["ctype"] == ic)[0]
            accepted = np.zeros(w["nobs"], bool)
            for rk2, s in zip(w["ranks"], st):
                accepted[rk2["gidx"][s["key"]]] = True
            sel = sel[accepted[sel]]
            x = g["ri"][sel] - w["ihalo"] - 0.5
            y = g["rj"][sel] - w["ihalo"] - 0.5
            owner_i = np.clip(np.ceil(x / w["nlon"]).astype(int) - 1, 0, px - 1)
            owner_j = np.clip(np.ceil(y / w["nlat"]).astype(int) - 1, 0, py - 1)
            ci = np.clip(np.ceil((x - owner_i * w["nlon"]) * gi / w["nlon"]).astype(int), 1, gi)
            cj = np.clip(np.ceil((y - owner_j * w["nlat"]) * gi / w["nlat"]).astype(int), 1, gj)
            xi = ci + (owner_i - rk["pi"]) * gi + si          # extended mesh index, 1-based
            xj = cj + (owner_j - rk["pj"]) * gj + sj
            inside = (xi >= 1) & (xi <= ei) & (xj >= 1) & (xj <= ej)
            for j in range(1, ej + 1):
                assert ac[j - 1, 0] == pos
                for i in range(1, ei + 1):
                    want = np.sort(sel[inside & (xi == i) & (xj == j)])
                    n = ac[j - 1, i] - ac[j - 1, i - 1]
                    assert n == len(want)
                    assert np.array_equal(got[pos:pos + n], want)
                    pos += n
        assert pos == nt


def test_monit_dep_and_additive_inflation_restatements():
    """row f4 restatements against numpy one-liners (common_obs_scale.f90:1851-1895, letkf_tools.f90:884-913)"""
    import ctypes as C
    import _oracle
    lib = _oracle.oracle()
    rng = np.random.default_rng(4)
    ids = np.array([2819, 3073, 3074, 4001, 4004, 4002], dtype=np.int32)
    nn = 5000
    elm = rng.choice(ids, nn).astype(np.int32)
    dep = rng.normal(0.5, 2.0, nn)
    qc = np.where(rng.random(nn) < 0.3, 5, 0).astype(np.int32)
    nobs, bias, rmse = np.zeros(6, np.int32), np.zeros(6), np.zeros(6)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    lib.orc_monit_dep(C.c_int(6), p(ids, C.c_int32), C.c_int64(nn), p(elm, C.c_int32), p(dep, C.c_double),
                      p(qc, C.c_int32), p(nobs, C.c_int32), p(bias, C.c_double), p(rmse, C.c_double))
    merged = np.where(elm == 3074, 3073, np.where(elm == 4004, 4001, elm))
    for u, i in enumerate(ids):
        sel = (merged == i) & (qc == 0)
        if i in (3074, 4004):
            assert nobs[u] == 0 and bias[u] == -9.99e33 and rmse[u] == -9.99e33
        else:
            assert nobs[u] == sel.sum()
            assert np.isclose(bias[u], dep[sel].mean(), rtol=1e-12) and np.isclose(rmse[u], np.sqrt((dep[sel] ** 2).mean()), rtol=1e-12)
    k, nv, nij1, nlev = 4, 7, 10, 3
    npts, nens = nij1 * nlev, k + 1
    anal = rng.normal(size=(nv, nens, npts))
    add = rng.normal(size=(nv, nens, npts))
    w = rng.uniform(0, 1, nij1)
    q = rng.uniform(1e-3, 1e-2, (nv, npts))
    sh = rng.permutation(k).astype(np.int32)
    want = anal.copy()
    for v in range(nv):
        for m in range(k):
            x = add[v, sh[m]] * 0.25 * np.tile(w, nlev)
            if 5 <= v <= 6:
                x = x * q[v]
            want[v, m] += x
    got = anal.copy()
    lib.orc_additive_inflation(C.c_int(k), C.c_int(nv), C.c_int64(npts), C.c_int64(nij1), p(got, C.c_double),
                               p(add, C.c_double), C.c_int64(1), C.c_int64(npts), C.c_int64(npts * nens),
                               C.c_double(0.25), p(w, C.c_double), p(q, C.c_double), C.c_int64(1), C.c_int64(npts),
                               C.c_int(5), C.c_int(6), p(sh, C.c_int32))
    assert np.allclose(got, want, rtol=1e-15, atol=0)


def test_h08_build_rules():
    """The -DH08 branches (letkf_obs.f90:432-469, :480-487, :520-541) written out row by row."""
    k, det = 12, True
    r = h08_rows(5, k, det, 3000)
    over = dict(h08=1, h08_min_cld_member=2, h08_limit_lev=20000.0, gross_error_h08=4.0, h08_bt_min=180.0)
    ens, val, qc, v2 = oracle_departure_h08(r, k, det, True, **over)
    seen = set()
    for n in range(len(qc)):
        e0, d = r["ens"][n], r["dat"][n]
        if r["qc"][n] > 0:
            assert qc[n] == r["qc"][n] and np.array_equal(ens[n], e0) and v2[n] == r["val2"][n]
            continue
        is_h08 = r["elm"][n] == 8800
        if is_h08 and (d == UNDEF or r["lev"][n] < 20000.0):
            assert qc[n] == 50 and np.array_equal(ens[n], e0) and v2[n] == r["val2"][n]
            seen.add("bad")
            continue
        m = e0[:k].copy()
        cld = 0
        if is_h08:
            cld = int((m < 0.0).sum())
            m = np.abs(m)
        mean = m[0]
        for i in range(1, k):
            mean = mean + m[i]
        mean = mean / k
        assert np.array_equal(ens[n, :k], m - mean) and val[n] == d - mean and ens[n, k] == d - e0[k]
        assert v2[n] == (abs(mean - r["val2"][n]) + abs(d - r["val2"][n])) * 0.5       # every live row of the build
        if is_h08:
            ge = 1.0 if cld < 2 else 4.0
            bad = abs(d - mean) > ge * r["err"][n] or d < 180.0
            seen.add(("clear" if cld < 2 else "cloudy") + ("_gross" if bad else "_good"))
        else:
            bad = abs(d - mean) > 5.0 * r["err"][n]
        assert qc[n] == (5 if bad else 0)
    assert seen == {"bad", "clear_gross", "clear_good", "cloudy_gross", "cloudy_good"}
