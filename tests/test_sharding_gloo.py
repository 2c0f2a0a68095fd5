"""The N>1 path on CPU: world_size-2 gloo run of the obs-table ALLGATHERV and the two point partitions
(scale-letkf_amd/sharding.py), the same functions bench.py uses with RCCL on the GPU box."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from __graft_entry__ import PKG_DIR, load_package


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(PKG_DIR))
    load_package()
    import importlib
    sh = importlib.import_module("scale_letkf_amd.sharding")
    try:
        # ragged shards: rank r owns 5 + 3r rows of a k+1 = 21 wide ensval table, plus an int32 column
        g = torch.Generator().manual_seed(100 + rank)
        rows = 5 + 3 * rank
        ens = torch.randn(rows, 21, generator=g, dtype=torch.float64)
        idx = torch.arange(rows, dtype=torch.int32) + 1000 * rank
        full, counts = sh.allgatherv_rows(ens)
        fidx, _ = sh.allgatherv_rows(idx)
        exp = torch.cat([torch.randn(5 + 3 * r, 21, generator=torch.Generator().manual_seed(100 + r),
                                     dtype=torch.float64) for r in range(world)])
        ok = torch.equal(full, exp) and counts == [5 + 3 * r for r in range(world)]
        ok = ok and fidx.tolist() == [i + 1000 * r for r in range(world) for i in range(5 + 3 * r)]
        # empty shard on one rank
        e2, c2 = sh.allgatherv_rows(ens[: (0 if rank == 0 else 4)])
        ok = ok and c2 == [0] + [4] * (world - 1) and e2.shape[0] == 4 * (world - 1)
        # partitions: every point exactly once
        nij = 37
        mine = sh.cyclic_points(nij, rank, world)
        allp = [torch.zeros(20, dtype=torch.int64) for _ in range(world)]
        pad = torch.full((20,), -1, dtype=torch.int64)
        pad[: mine.numel()] = mine
        dist.all_gather(allp, pad)
        got = sorted(int(v) for t in allp for v in t.tolist() if v >= 0)
        ok = ok and got == list(range(nij))
        tiles = sh.tile_partition(10, 7, world)
        cover = torch.zeros(7, 10, dtype=torch.int64)
        for (i0, i1, j0, j1) in tiles:
            cover[j0:j1, i0:i1] += 1
        ok = ok and bool((cover == 1).all())
        # halo filter
        ri = torch.tensor([0.5, 4.9, 5.1, 9.5])
        rj = torch.tensor([1.0, 1.0, 6.0, 6.5])
        keep = sh.halo_rows(ri, rj, tiles[rank], 0.5, 0.5)
        ok = ok and keep.numel() >= 1
        # pairwise exchange of ragged blocks (the halo-only alternative): rank r sends 2 + r + 3q rows to rank q, none to
        # itself when r == 1; every received row must be the sender's
        def block(src, dst):
            n = 0 if (src == dst == 1) else 2 + src + 3 * dst
            return (torch.arange(n * 4, dtype=torch.float64).reshape(n, 4) + 1000.0 * src + 100.0 * dst)
        got = sh.exchange_rows([block(rank, d) for d in range(world)], [block(s, rank).shape[0] for s in range(world)])
        ok = ok and all(torch.equal(got[s], block(s, rank)) for s in range(world))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
