"""The member-major <-> point-major transpose across ranks (scatter / gather_grd_mpi_alltoall,
scale/common/common_mpi_scale.f90:1279-1396): sharding.scatter_members_alltoall / gather_members_alltoall on 2 and 3
ranks that share the test box's one GPU and exchange through gloo (RCCL refuses two ranks on one device), the device
side being the library's letkf_member_points_dev.  Checked bit for bit against the definition: point i of rank p is
subdomain point p + np i (grd_to_buf), every member in its slot; and gather(scatter(field)) == field."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from __graft_entry__ import PKG_DIR, load_package

pytestmark = pytest.mark.gpu

NLEV, NLON, NLAT, NV = 9, 13, 7, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def field(m):
    return np.random.default_rng(1000 + m).standard_normal(NV * NLAT * NLON * NLEV)   # v3dg(nlev,nlon,nlat,nv3d), level-fastest


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(PKG_DIR))
    pkg = load_package()
    import importlib
    sh = importlib.import_module("scale_letkf_amd.sharding")
    try:
        dev = torch.device("cuda:0")
        ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
        nens = 2 * world                                   # members 0 .. 2 world - 2 arrive in two batches, the last one short
        nij1 = sh.nij1_of(NLON, NLAT, rank, world)
        x = torch.full((NV * nens * NLEV * nij1,), float("nan"), dtype=torch.float64, device=dev)
        batches = [(0, world), (world, world - 1)]
        for mstart, mcount in batches:
            mine = torch.from_numpy(field(mstart + rank)).to(dev) if rank < mcount else None
            sh.scatter_members_alltoall(ctx, NLEV, NLON, NLAT, NV, mstart, mcount, mine, x, nens)
        torch.cuda.synchronize()
        xv = x.cpu().numpy().reshape(NV, nens, NLEV, nij1)
        ok = True
        for m in range(2 * world - 1):
            f = field(m).reshape(NV, NLAT * NLON, NLEV)
            for i in range(nij1):
                ok = ok and np.array_equal(xv[:, m, :, i], f[:, rank + world * i, :])
        ok = ok and np.isnan(xv[:, 2 * world - 1]).all()   # the slot nobody sent stays untouched
        # and back
        for mstart, mcount in batches:
            back = torch.zeros(NV * NLAT * NLON * NLEV, dtype=torch.float64, device=dev) if rank < mcount else None
            sh.gather_members_alltoall(ctx, NLEV, NLON, NLAT, NV, mstart, mcount, x, nens, back)
            if rank < mcount:
                torch.cuda.synchronize()
                ok = ok and np.array_equal(back.cpu().numpy(), field(mstart + rank))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_alltoall_transpose(world):
    port = _free_port()
    c = mp.get_context("spawn")
    q = c.Queue()
    procs = [c.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
