"""2-D variables (nv2d > 0) at ilev = 1, scale/letkf/letkf_tools.f90:530-659 inside the loop body :313-686.

The reference updates the 2-D variables of a horizontal point right after its level-1 3-D variables, sharing the
weights of a variable-localisation class between the two loops (trans_done, :544-568).  Through the C ABI that is the
SAME entry, letkf_das_points_dev, on a variable list that holds the 2-D variables next to 3-D ones -- element (p, m, v)
is addressed by strides, so the caller hands in an assembled level-1 array ("2-D variables"):
  (i)  generic: all nv3d + nv2d variables, one call per class with var_mask (any nv: the staged path);
  (ii) fast, for the reference's nv3d = 11: the 3-D call as always, plus ONE 11-slot array [class representatives,
       2-D variables, padding] per call -- the one-wave kernel; the representatives' analyses are discarded.
Both must reproduce the oracle's restatement of the reference's two loops (oracle/letkf_oracle.c,
orc_das_letkf_level1_2d) to 1e-10 * max(|mean|, |x'|), inflation slots to 1e-12.
"""
import numpy as np
import pytest
import torch

import _oracle
from _cases import das_case

pytestmark = pytest.mark.gpu

NV3, NV2 = 11, 3
# 1-based, as var_local_n2nc / var_local_n2n (letkf_tools.f90:139-157): 3-D variables 1-5 class 1, 6-11 class 2;
# 2-D variable 12 shares class 1 (representative: 3-D variable 1), 13 and 14 form class 3 (representative 13, a 2-D one)
N2NC = [1] * 5 + [2] * 6 + [1, 3, 3]
N2N = [1] * 5 + [6] * 6 + [1, 13, 13]
VARLOC = {1: 1.0, 2: 0.6, 3: 0.8}

CONFIGS = {
    "rtps_adaptive_det": dict(det_run=1, infl_adaptive=1, relax_alpha_spread=0.95, relax_to_inflated_prior=1),
    "rtpp_qtop_qclamp": dict(relax_alpha=0.8, q_update_top=1e9, q_sprd_max=0.5),   # every moisture variable skipped
    "none": dict(infl_adaptive=1),
}


def make(k, nij1, seed, det):
    c = das_case(k=k, nv=NV3 + NV2, npts=nij1, nobs_tot=400, n_mean=70, seed=seed, det_run=det, infl0=1.05)
    nens = c["nens"]
    x = c["gues"].reshape(NV3 + NV2, nens, nij1)
    gues3 = np.ascontiguousarray(x[:NV3]).reshape(-1)          # gues3d(ij, ilev = 1, m, n): sp 1, sm nij1, sv nij1 nens
    gues2 = np.ascontiguousarray(x[NV3:]).reshape(-1)          # gues2d(ij, m, n)
    infl = c["infl"] * (1.0 + 0.02 * np.arange(c["infl"].size) / c["infl"].size)
    work3d, work2d = infl[:nij1 * NV3].copy(), infl[nij1 * NV3:].copy()
    lists = {cl: (c["rdiag"] / f, c["rloc"] * f) for cl, f in VARLOC.items()}
    return c, gues3, gues2, work3d, work2d, lists


def oracle_run(c, cfg, gues3, gues2, work3d, work2d, lists, k, nij1, det):
    prm = _oracle.DasParams(k=k, nv=NV3, det_run=int(det), infl_adaptive=cfg.get("infl_adaptive", 0),
                            relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                            relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                            q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0), iv_p=4,
                            iv_q_first=5, iv_q_last=10, nthreads=1, var_mask=0)
    nnz = c["obs_idx"].size
    off = np.stack([c["obs_off"] + i * nnz for i in range(3)])
    idx = np.concatenate([c["obs_idx"]] * 3)
    rd = np.concatenate([lists[cl][0] for cl in (1, 2, 3)])
    rl = np.concatenate([lists[cl][1] for cl in (1, 2, 3)])
    s = (1, nij1, nij1 * c["nens"])
    r = _oracle.das_level1_2d(prm, NV2, N2NC, N2N, 3, nij1, off, idx, rd, rl, c["ensval"], c["dep"], c["beta"], work3d,
                              work2d, gues3, s, gues2, s)
    assert r["rc"] == 0
    return r


def check(ref, anal3, anal2, w3, w2, gues3, gues2, k, nens, nij1, det):
    for name, got, want, gues, nv in (("3d", anal3, ref["anal3"], gues3, NV3), ("2d", anal2, ref["anal2"], gues2, NV2)):
        g, e, x = got.reshape(nv, nens, nij1), want.reshape(nv, nens, nij1), gues.reshape(nv, nens, nij1)
        rows = list(range(k)) + ([k + 1] if det else [])
        for v in range(nv):
            scale = np.maximum(np.abs(x[v, k]), np.abs(x[v, :k]).max(axis=0))
            err = np.abs(g[v, rows] - e[v, rows]).max(axis=0) / scale
            assert err.max() <= 1e-10, (name, v, err.max())
    assert np.abs(w3 - ref["work3d"]).max() <= 1e-12
    assert np.abs(w2 - ref["work2d"]).max() <= 1e-12


def das_call(cfg, k, nv, c, lists_cl, infl, gues, anal, nij1, det, mask, q_on=True):
    from _gpu import ctx, dev
    status = torch.full((nij1,), -1, dtype=torch.int32, device="cuda")
    ctx().das_points(k, nv, dev(c["obs_off"]), dev(c["obs_idx"]), dev(lists_cl[0]), dev(lists_cl[1]), dev(c["ensval"]),
                     c["kld"], dev(c["dep"]), infl, gues, anal, 1, nij1, nij1 * c["nens"], beta=dev(c["beta"]), det_run=det,
                     infl_adaptive=cfg.get("infl_adaptive", 0), relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                     relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                     q_update_top=cfg.get("q_update_top", 0.0) if q_on else 0.0,
                     q_sprd_max=cfg.get("q_sprd_max", 0.0) if q_on else 0.0, iv_p=4, iv_q_first=5,
                     iv_q_last=10 if q_on else 4, status=status, var_mask=mask)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()


@pytest.mark.parametrize("k", [20, 50])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_level1_with_2d_variables_generic(name, k):
    """(i): one assembled array of nv3d + nv2d variables, one call per class"""
    from _gpu import dev
    cfg = CONFIGS[name]
    det = bool(cfg.get("det_run", 0))
    nij1 = 18
    c, gues3, gues2, work3d, work2d, lists = make(k, nij1, 900 + k, det)
    ref = oracle_run(c, cfg, gues3, gues2, work3d, work2d, lists, k, nij1, det)
    nv = NV3 + NV2
    gues = dev(np.concatenate([gues3, gues2]))
    anal = torch.full_like(gues, float("nan"))
    infl = dev(np.concatenate([work3d, work2d]))
    for cl in (1, 2, 3):
        mask = sum(1 << v for v in range(nv) if N2NC[v] == cl)
        das_call(cfg, k, nv, c, lists[cl], infl, gues, anal, nij1, det, mask)
    a, w = anal.cpu().numpy(), infl.cpu().numpy()
    n3 = gues3.size
    check(ref, a[:n3], a[n3:], w[:nij1 * NV3], w[nij1 * NV3:], gues3, gues2, k, c["nens"], nij1, det)


@pytest.mark.parametrize("k", [20, 50, 100])
@pytest.mark.parametrize("name", ["rtps_adaptive_det", "none"])
def test_level1_with_2d_variables_fast_path(name, k):
    """(ii): the 3-D call on gues3d as always, the 2-D variables through an 11-slot array on the register kernels.
    The 2-D call runs on the PRIOR inflation of the representatives (a copy taken before the 3-D call updates it)."""
    from _gpu import ctx, dev
    cfg = CONFIGS[name]
    det = bool(cfg.get("det_run", 0))
    nij1 = 18
    c, gues3, gues2, work3d, work2d, lists = make(k, nij1, 950 + k, det)
    ref = oracle_run(c, cfg, gues3, gues2, work3d, work2d, lists, k, nij1, det)
    nens = c["nens"]
    # ---- 2-D call first: slots [3-D variable 1 (representative of class 1), 2-D 12, 13, 14, 7 x padding]
    x3, x2 = gues3.reshape(NV3, nens, nij1), gues2.reshape(NV2, nens, nij1)
    packed = np.zeros((11, nens, nij1))
    packed[0], packed[1:4] = x3[0], x2
    pinfl = np.ones((11, nij1))
    pinfl[0], pinfl[1:4] = work3d.reshape(NV3, nij1)[0], work2d.reshape(NV2, nij1)
    g2, a2, i2 = dev(packed.reshape(-1)), torch.full((packed.size,), float("nan"), dtype=torch.float64, device="cuda"), dev(pinfl.reshape(-1))
    das_call(cfg, k, 11, c, lists[1], i2, g2, a2, nij1, det, 0b0011, q_on=False)    # class 1: representative + variable 12
    das_call(cfg, k, 11, c, lists[3], i2, g2, a2, nij1, det, 0b1100, q_on=False)    # class 3: variables 13, 14
    # (k = 100: the loop body without k x k outputs takes the staged path's eigen-free route since r2, tests/test_gpu_poly.py)
    assert ctx().last_path().startswith("letkf_trio_kernel" if k <= 20 else "letkf_wave_kernel" if k < 63 else "staged")
    # ---- the 3-D call, one per class
    g3, a3, i3 = dev(gues3), torch.full((gues3.size,), float("nan"), dtype=torch.float64, device="cuda"), dev(work3d)
    for cl in (1, 2):
        das_call(cfg, k, 11, c, lists[cl], i3, g3, a3, nij1, det, sum(1 << v for v in range(NV3) if N2NC[v] == cl))
    anal2 = a2.cpu().numpy().reshape(11, nens, nij1)[1:4].reshape(-1)
    w2 = i2.cpu().numpy().reshape(11, nij1)[1:4].reshape(-1)
    check(ref, a3.cpu().numpy(), anal2, i3.cpu().numpy(), w2, gues3, gues2, k, nens, nij1, det)
