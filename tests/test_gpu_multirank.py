"""The N > 1 paths of bench.py on real kernels, rehearsed on the one GPU a test box has: the ranks share the device and
exchange through gloo (LETKF_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device), everything else -- mesh sort,
the all-gather of the sorted observation buffers and cell counts, the extended-subdomain plan with its halo, obs_local
and the loop body per tile -- is the code the 8-GPU run executes.  Checked: every rank finishes, no point reports a
status, and the tiled domain finds exactly the local observations of the single domain (same mean list length)."""
import json
import os
import subprocess
import sys

import pytest

from __graft_entry__ import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*extra):
    env = dict(os.environ, LETKF_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C2-mini", "--steps", "1",
                          "--warmup", "1", "--no-cpu-baseline", *extra], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_strong_scaling_tiles_two_and_four_ranks():
    one = run_bench("--scaling", "strong")
    n1 = float(one["config"]["workload"].split("mean ")[1].split(" ")[0])
    for n in (2, 4):
        d = run_bench("--gpus", str(n), "--scaling", "strong")
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["nonzero_status_points"] == 0
        nn = float(d["config"]["workload"].split("mean ")[1].split(" ")[0])
        assert nn == n1, (nn, n1)                       # the halo plan hands every tile all it needs, nothing twice
        assert d["config"]["points_total"] == one["config"]["points_total"]
        assert d["config"]["obs_rows_per_rank_with_halo"] < one["config"]["obs_rows_per_rank_with_halo"]


def test_weak_scaling_two_ranks():
    d = run_bench("--gpus", "2")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["nonzero_status_points"] == 0
    assert d["value"] > 0 and "all-gather" in d["config"]["parallelism"]


def test_halo_only_exchange_gives_the_all_gather_analysis():
    """--exchange halo: every rank receives only the rows of its extended subdomain (pairwise sends, sharding.exchange_rows)
    and must end up with the very table -- hence the very analysis -- the ALLGATHERV + halo plan gives."""
    for n in (2, 4):
        a = run_bench("--gpus", str(n), "--scaling", "strong")
        h = run_bench("--gpus", str(n), "--scaling", "strong", "--exchange", "halo")
        assert h["nonzero_status_points"] == 0
        assert h["config"]["obs_rows_per_rank_with_halo"] == a["config"]["obs_rows_per_rank_with_halo"]
        assert h["config"]["obs_rows_received_per_rank"] < a["config"]["obs_rows_received_per_rank"]
        assert abs(h["anal_checksum"] - a["anal_checksum"]) <= 1e-12 * abs(a["anal_checksum"]), (h["anal_checksum"], a["anal_checksum"])


def test_configs3_tiling_rehearsed_with_two_and_four_ranks():
    """`bench.py --gpus 8 --scaling strong --workload C4 --lists pipeline` is the line for BASELINE configs[3] the day an 8-GPU
    node runs it (4 x 2 tiles of the C4-gpu tile's size, one call of letkf_das_columns_dev per tile and step).  Rehearsed here on
    a small domain of configs[3]'s observation density with the ranks a one-GPU box admits on its card beside the test process
    (the process guard stops at 6 processes: a 2 x 1 and a 2 x 2 tiling): same local observations as the single domain, every
    status 0."""
    import bench_workload as bw
    assert bw.CONFIGS["C4"]["nx"] == 1000 and bw.CONFIGS["C4"]["nx"] // 4 == bw.CONFIGS["C4-gpu"]["nx"] and bw.CONFIGS["C4"]["ny"] // 2 == bw.CONFIGS["C4-gpu"]["ny"]

    def run(*extra):
        env = dict(os.environ, LETKF_BENCH_BACKEND="gloo")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C4-dom-mini", "--steps", "1", "--warmup", "0",
                              "--no-cpu-baseline", "--scaling", "strong", "--lists", "pipeline", "--list-gb", "0.05", *extra], env=env,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    one = run()
    n1 = float(one["config"]["workload"].split("mean ")[1].split(" ")[0])
    assert n1 > 2000
    for n, tiles in ((2, "2x1 tiles"), (4, "2x2 tiles")):
        d = run("--gpus", str(n))
        nn = float(d["config"]["workload"].split("mean ")[1].split(" ")[0])
        assert nn == n1, (n, nn, n1)
        assert d["n_gpus"] == n and d["nonzero_status_points"] == 0 and tiles in d["config"]["workload"]
        assert "letkf_das_columns_dev" in d["config"]["workload"]
        assert d["config"]["points_total"] == one["config"]["points_total"] == 48 * 32 * 6
        # (every rank draws its own state: the checksums of different tilings are not comparable; that a tiled analysis IS the
        # single-domain one is tests/test_gpu_tiles.py's)
