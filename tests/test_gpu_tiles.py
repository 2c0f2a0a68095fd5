"""A sharded analysis equals the unsharded one (SURVEY.md section 8 row e).  One domain, analysed
(a) as a single subdomain and (b) cut into 2 x 2 and 3 x 2 tiles whose ranks each run the set_letkf_obs pipeline --
departure + QC, mesh sort, all-gather (a concatenation between virtual ranks here; RCCL in bench.py), extended-subdomain
plan with the localisation halo (scale/letkf/letkf_obs.f90:922-976, 1036-1109), row gathers -- then obs_local and the
loop body for their own grid points.  Required: every grid point gets the SAME local observations in the SAME order
with bit-identical localisation weights, and the stitched analysis members agree to 1e-10 (the warm-started eigensolves
run along different point runs in the two decompositions, so not bit for bit).
Non-square subdomains (3 x 2 tiles of 8 x 12 points): the reference's ij_obsgrd scales rj with ngrd_i
(letkf_obs.f90:1200) while the lookup ij_obsgrd_ext uses ngrd_j (:1223) -- SURVEY.md 9.10.  Restated as written,
so on such tiles the sort and the lookup disagree and a point can LOSE observations that sit in a mesh row the
lookup does not visit, exactly as in the reference; test_non_square_tiles_follow_the_reference_quirk pins that
behaviour (every tiled list is a subset of the single-domain list, with the same weights)."""
import numpy as np
import pytest

from _tiles import tiled_analysis

pytestmark = pytest.mark.gpu


def first_guess(seed, nv, k, det_run, nlev, nlat, nlon):
    rng = np.random.default_rng(seed)
    nens = k + 1 + (1 if det_run else 0)
    x = rng.standard_normal((nv, nens, nlev, nlat, nlon))
    x[:, :k] *= np.array([2.0, 2.0, 2.0, 1.0, 50.0] + [1e-3] * (nv - 5))[:, None, None, None, None]
    x[:, :k] -= x[:, :k].mean(axis=1, keepdims=True)
    x[:, k] = (rng.standard_normal((nv, nlev, nlat, nlon)) * 5.0 + 50.0)
    x[5:, k] = np.abs(x[5:, k]) * 1e-3
    if det_run:
        x[:, k + 1] = x[:, k] + rng.standard_normal((nv, nlev, nlat, nlon))
    return x


@pytest.mark.parametrize("px,py,k", [(2, 2, 20), (3, 3, 50)])
def test_tiled_analysis_equals_single_domain(px, py, k):
    nlon_g, nlat_g, nlev, nv, nobs = 24, 24, 3, 11, 1500
    zlev = np.array([800.0, 5000.0, 9500.0])
    x = first_guess(40 + k, nv, k, True, nlev, nlat_g, nlon_g)
    one = tiled_analysis(77, 1, 1, nlon_g, nlat_g, nlev, k, nobs, x, zlev)
    til = tiled_analysis(77, px, py, nlon_g, nlat_g, nlev, k, nobs, x, zlev)
    # the halo plan gives every rank fewer rows than the whole table, and more than its own subdomain holds
    assert all(n < one["nrows"][0] for n in til["nrows"])
    assert sum(til["nrows"]) > one["nrows"][0]
    # local lists: same observations, same order, same weights
    ntot = 0
    for key, (gid1, rd1, rl1) in one["lists"].items():
        gid2, rd2, rl2 = til["lists"][key]
        assert np.array_equal(gid1, gid2), key
        assert np.array_equal(rd1, rd2) and np.array_equal(rl1, rl2), key
        ntot += len(gid1)
    assert ntot / len(one["lists"]) > 50
    # analysis members and the deterministic member
    a1, a2 = one["anal"], til["anal"]
    members = list(range(k)) + [k + 1]
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        err = np.abs(a1[v, members] - a2[v, members]).max()
        assert np.isfinite(a2[v, members]).all()
        assert err <= 1e-10 * scale, (v, err, scale)
    # and the analysis did something
    assert np.abs(a1[0, :k] - (x[0, :k] + x[0, k:k + 1])).max() > 1e-3


def test_non_square_tiles_with_the_ij_obsgrd_fix():
    """letkf_mesh.fix_ij_obsgrd = 1 (rj scaled with ngrd_j in the sort, as the lookup does): 3 x 2 tiles of 8 x 12
    points give the single domain's lists (as sets: the mesh rows differ from the square case) and analysis."""
    nlon_g, nlat_g, nlev, nv, nobs, k = 24, 24, 2, 11, 1500, 20
    zlev = np.array([800.0, 9500.0])
    x = first_guess(8, nv, k, True, nlev, nlat_g, nlon_g)
    one = tiled_analysis(79, 1, 1, nlon_g, nlat_g, nlev, k, nobs, x, zlev)
    til = tiled_analysis(79, 3, 2, nlon_g, nlat_g, nlev, k, nobs, x, zlev, fix_ij_obsgrd=True)
    for key, (gid1, rd1, rl1) in one["lists"].items():
        gid2, rd2, rl2 = til["lists"][key]
        o1, o2 = np.argsort(gid1, kind="stable"), np.argsort(gid2, kind="stable")
        assert np.array_equal(gid1[o1], gid2[o2]), key
        assert np.array_equal(rd1[o1], rd2[o2]) and np.array_equal(rl1[o1], rl2[o2]), key
    members = list(range(k)) + [k + 1]
    for v in range(nv):
        scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
        assert np.abs(one["anal"][v, members] - til["anal"][v, members]).max() <= 1e-10 * scale


def test_non_square_tiles_follow_the_reference_quirk():
    nlon_g, nlat_g, nlev, nv, nobs, k = 24, 24, 2, 11, 1500, 20
    zlev = np.array([800.0, 9500.0])
    x = first_guess(7, nv, k, True, nlev, nlat_g, nlon_g)
    one = tiled_analysis(78, 1, 1, nlon_g, nlat_g, nlev, k, nobs, x, zlev)
    til = tiled_analysis(78, 3, 2, nlon_g, nlat_g, nlev, k, nobs, x, zlev)
    n1 = n2 = 0
    for key, (gid1, rd1, rl1) in one["lists"].items():
        gid2, rd2, rl2 = til["lists"][key]
        w1 = dict(zip(gid1.tolist(), zip(rd1.tolist(), rl1.tolist())))
        for g, rd, rl in zip(gid2.tolist(), rd2.tolist(), rl2.tolist()):
            assert w1[g] == (rd, rl), key          # what is found is found with the same weights
        n1 += len(gid1)
        n2 += len(gid2)
    assert n2 <= n1 and n2 > 0.9 * n1
