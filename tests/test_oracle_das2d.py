"""CPU: the oracle's restatement of the level-1 loop body with 2-D variables (letkf_tools.f90:313-686,
orc_das_letkf_level1_2d: both loops, shared trans_done) against its single-class restatement (orc_das_letkf_points,
:313-527) run on the assembled variable list, one pass per class -- the equivalence INTEGRATION.md's recipe rests on.
Same arithmetic in the same order: bit-identical."""
import numpy as np
import pytest

import _oracle
from test_gpu_das2d import CONFIGS, N2NC, NV2, NV3, make, oracle_run


@pytest.mark.parametrize("name", list(CONFIGS))
def test_two_loop_restatement_equals_per_class_passes(name):
    cfg = CONFIGS[name]
    det = bool(cfg.get("det_run", 0))
    k, nij1 = 12, 14
    c, gues3, gues2, work3d, work2d, lists = make(k, nij1, 321, det)
    ref = oracle_run(c, cfg, gues3, gues2, work3d, work2d, lists, k, nij1, det)
    nv = NV3 + NV2
    gues = np.concatenate([gues3, gues2])
    infl = np.concatenate([work3d, work2d])
    anal = np.full_like(gues, np.nan)
    for cl in (1, 2, 3):
        mask = sum(1 << v for v in range(nv) if N2NC[v] == cl)
        prm = _oracle.DasParams(k=k, nv=nv, det_run=int(det), infl_adaptive=cfg.get("infl_adaptive", 0),
                                relax_to_inflated_prior=cfg.get("relax_to_inflated_prior", 0),
                                relax_alpha=cfg.get("relax_alpha", 0.0), relax_alpha_spread=cfg.get("relax_alpha_spread", 0.0),
                                q_update_top=cfg.get("q_update_top", 0.0), q_sprd_max=cfg.get("q_sprd_max", 0.0), iv_p=4,
                                iv_q_first=5, iv_q_last=10, nthreads=1, var_mask=mask)
        r = _oracle.das_points(prm, c["obs_off"], c["obs_idx"], lists[cl][0], lists[cl][1], c["ensval"], c["dep"],
                               c["beta"], infl, gues, 1, nij1, nij1 * c["nens"])
        assert r["rc"] == 0
        sel = np.array([(mask >> v) & 1 for v in range(nv)], bool)
        anal.reshape(nv, -1)[sel] = r["anal"].reshape(nv, -1)[sel]
        infl = r["infl"]
    nens = c["nens"]
    rows = list(range(k)) + ([k + 1] if det else [])
    a = anal.reshape(nv, nens, nij1)[:, rows]
    want = np.concatenate([ref["anal3"], ref["anal2"]]).reshape(nv, nens, nij1)[:, rows]
    assert np.array_equal(a, want)
    assert np.array_equal(infl, np.concatenate([ref["work3d"], ref["work2d"]]))
