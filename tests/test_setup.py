"""das_letkf's set-up tables (scale/letkf/letkf_tools.f90:130-267): the oracle's
restatement against independent formulations, and the C-ABI host helpers against the oracle -- integers, bit exact.
The host helpers need no device (they derive tens of integers once per analysis)."""
import ctypes as C

import numpy as np
import pytest

import _oracle
from __graft_entry__ import load_package

pkg = load_package()


def orc_classes(var_local):
    v = np.asfortranarray(var_local, dtype=np.float64)
    nvar, nlt = v.shape
    n2nc = np.zeros(nvar, dtype=np.int32)
    n2n = np.zeros(nvar, dtype=np.int32)
    nc = C.c_int32(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _oracle.oracle().orc_var_local_classes(C.c_int(nvar), C.c_int(nlt), p(v), p(n2nc), p(n2n), C.byref(nc))
    return n2nc, n2n, nc.value


def orc_merge(elm_u, typ, cm):
    eu = np.ascontiguousarray(elm_u, dtype=np.int32)
    ty = np.ascontiguousarray(typ, dtype=np.int32)
    cmf = np.asfortranarray(cm, dtype=np.int32)
    nct = len(eu)
    n_merge = np.zeros(nct, dtype=np.int32)
    ic_merge = np.full((nct, nct), -1, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _oracle.oracle().orc_ctype_merge(C.c_int(nct), p(eu), p(ty), C.c_int(cmf.shape[0]), C.c_int(cmf.shape[1]), p(cmf),
                                     p(n_merge), p(ic_merge))
    return n_merge, ic_merge


def groups_from_oracle(n_merge, ic_merge):
    gs, gm = [0], []
    for ic in range(len(n_merge)):
        if n_merge[ic] > 0:
            gm += list(ic_merge[ic, :n_merge[ic]])
            gs.append(len(gm))
    return np.array(gs, dtype=np.int32), np.array(gm, dtype=np.int32)


def test_classes_default_namelist_is_one_class():
    # common_nml.f90:221-229: every VAR_LOCAL_* defaults to 1.0 -> a single class, one solve per point
    n2nc, n2n, nc = orc_classes(np.ones((11, 9)))
    assert nc == 1 and (n2nc == 0).all() and (n2n == 0).all()
    g = pkg.var_local_classes(np.ones((11, 9)))
    assert g[2] == 1 and (g[0] == 0).all() and (g[1] == 0).all()


def test_classes_against_unique_rows():
    """when the distinct rows come first, the classes are the unique rows in order of first appearance"""
    rng = np.random.default_rng(3)
    for trial in range(20):
        nc = int(rng.integers(1, 5))
        rows = rng.choice([0.0, 0.5, 1.0], size=(nc, 9))
        rows = np.unique(rows, axis=0)
        rng.shuffle(rows)
        nc = len(rows)
        pick = np.concatenate([np.arange(nc), rng.integers(0, nc, size=11 - nc)])
        vl = rows[pick]
        n2nc, n2n, got_nc = orc_classes(vl)
        assert got_nc == nc
        assert (n2nc == pick).all()
        assert (n2n == pick).all()          # representative of class c is variable c here
        g = pkg.var_local_classes(vl)
        assert g[2] == got_nc and (g[0] == n2nc).all() and (g[1] == n2n).all()


def test_classes_reference_quirk_restated_as_written():
    """letkf_tools.f90:145 indexes var_local with the class NUMBER of variable i: rows (a, a, b, b) give three classes"""
    a, b = np.ones(9), np.full(9, 0.5)
    vl = np.stack([a, a, b, b])
    n2nc, n2n, nc = orc_classes(vl)
    assert nc == 3 and list(n2nc) == [0, 0, 1, 2] and list(n2n) == [0, 0, 2, 3]
    g = pkg.var_local_classes(vl)
    assert g[2] == 3 and list(g[0]) == [0, 0, 1, 2] and list(g[1]) == [0, 0, 2, 3]


def test_classes_random_bit_exact_vs_oracle():
    rng = np.random.default_rng(11)
    for trial in range(200):
        nvar = int(rng.integers(1, 14))
        vl = rng.choice([0.0, 1.0, 0.3], size=(nvar, 9), p=[0.2, 0.6, 0.2])
        if trial % 3 == 0:
            vl[:] = vl[rng.integers(0, nvar, size=nvar)]
        o = orc_classes(vl)
        g = pkg.var_local_classes(vl)
        assert g[2] == o[2] and (g[0] == o[0]).all() and (g[1] == o[1]).all()


def test_merge_groups_radar_ref_and_zero():
    """the reference's one merge class: reflectivity (uid 9) and zero-reflectivity (uid 10) of type 22 (:170-171)"""
    nid, nobt = 16, 24
    cm = np.zeros((nid, nobt), dtype=np.int32)
    cm[9 - 1, 22 - 1] = 1
    cm[10 - 1, 22 - 1] = 1
    #            T/ADPUPA  REF/22  Vr/22   RE0/22  PS/ADPSFC
    elm_u = [3, 9, 11, 10, 7]
    typ = [1, 22, 22, 22, 8]
    n_merge, ic_merge = orc_merge(elm_u, typ, cm)
    assert list(n_merge) == [1, 2, 1, 0, 1]
    assert list(ic_merge[1, :2]) == [1, 3]
    gs, gm = pkg.ctype_merge_groups(elm_u, typ, cm)
    egs, egm = groups_from_oracle(n_merge, ic_merge)
    assert (gs == egs).all() and (gm == egm).all()
    assert list(gs) == [0, 1, 3, 4, 5] and list(gm) == [0, 1, 3, 2, 4]


def test_merge_groups_random_bit_exact_vs_oracle():
    rng = np.random.default_rng(5)
    nid, nobt = 16, 24
    for trial in range(200):
        cm = np.zeros((nid, nobt), dtype=np.int32)
        for _ in range(int(rng.integers(0, 12))):
            cm[rng.integers(0, nid), rng.integers(0, nobt)] = rng.integers(1, 4)
        nct = int(rng.integers(0, 12))
        elm_u = rng.integers(1, nid + 1, size=nct)
        typ = rng.integers(1, nobt + 1, size=nct)
        if nct and trial % 2:
            hot = np.argwhere(cm > 0)
            for i in range(nct):
                if len(hot) and rng.random() < 0.6:
                    e, t = hot[rng.integers(0, len(hot))]
                    elm_u[i], typ[i] = e + 1, t + 1
        n_merge, ic_merge = orc_merge(elm_u, typ, cm)
        egs, egm = groups_from_oracle(n_merge, ic_merge)
        gs, gm = pkg.ctype_merge_groups(elm_u, typ, cm)
        assert (gs == egs).all() and (gm[:len(egm)] == egm).all()
        # independent formulation: a partition of the ctypes; positive merge classes sit together, master = smallest
        assert sorted(gm[:nct]) == list(range(nct))
        for g in range(len(gs) - 1):
            mem = gm[gs[g]:gs[g + 1]]
            cls = {int(cm[elm_u[i] - 1, typ[i] - 1]) for i in mem}
            assert len(cls) == 1 and mem[0] == min(mem) and (len(mem) == 1 or cls.pop() > 0)


def test_radar_only():
    o = _oracle.oracle()
    for typ in ([22, 22, 22], [22, 1, 22], [1], []):
        ty = np.ascontiguousarray(typ, dtype=np.int32)
        want = int(all(t == 22 for t in typ))
        assert o.orc_radar_only(C.c_int(len(typ)), ty.ctypes.data_as(C.c_void_p), C.c_int(22)) == want
        assert pkg.radar_only(typ) == want


def test_infl_init_oracle():
    o = _oracle.oracle()
    w = np.linspace(0.5, 1.5, 11)
    a = w.copy()
    o.orc_infl_init(C.c_int64(11), a.ctypes.data_as(C.c_void_p), C.c_double(1.2), C.c_double(0.0))
    assert (a == 1.2).all()
    a = w.copy()
    o.orc_infl_init(C.c_int64(11), a.ctypes.data_as(C.c_void_p), C.c_double(-1.0), C.c_double(0.9))
    assert (a == np.maximum(w, 0.9)).all()
    a = w.copy()
    o.orc_infl_init(C.c_int64(11), a.ctypes.data_as(C.c_void_p), C.c_double(0.8), C.c_double(0.9))
    assert (a == 0.9).all()


def test_relax_beta_oracle_against_formula():
    """orc_relax_beta (letkf_tools.f90:1911-1948) against a numpy statement of the same taper"""
    class BP(C.Structure):
        _fields_ = [("radar_only", C.c_int), ("radar_zmax", C.c_double), ("vert_local_radar", C.c_double),
                    ("boundary_buffer_width", C.c_double), ("dx", C.c_double), ("dy", C.c_double), ("ihalo", C.c_int),
                    ("jhalo", C.c_int), ("nlong", C.c_int), ("nlatg", C.c_int)]
    o = _oracle.oracle()
    rng = np.random.default_rng(2)
    bp = BP(1, 10000.0, 2000.0, 5000.0, 1000.0, 1000.0, 2, 2, 40, 32)
    dzf = float(np.float32(3.651483717))
    for _ in range(500):
        ri, rj, rz = rng.uniform(2.0, 43.0), rng.uniform(2.0, 35.0), rng.uniform(0.0, 25000.0)
        got = o.orc_relax_beta(C.byref(bp), C.c_double(ri), C.c_double(rj), C.c_double(rz))
        if rz > 10000.0 + 2000.0 * dzf:
            want = 0.0
        else:
            d = min(min(ri - 2, 40 + 2 + 1 - ri) * 1000.0, min(rj - 2, 32 + 2 + 1 - rj) * 1000.0) / 5000.0
            want = max(d, 0.0) if d < 1.0 else 1.0
        assert got == want
