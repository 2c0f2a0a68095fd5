"""Grid-point sharding across the GPUs of one node and the one exchange the path has.

The analysis is embarrassingly parallel over grid points (scale/letkf/letkf_tools.f90:313-686 has no communication
inside the loop); what the reference exchanges BEFORE the loop is the QC-passed observation table, with
MPI_ALLGATHERV over the subdomain communicator (scale/letkf/letkf_obs.f90:1036-1046), after which every rank keeps
the rows that fall into its extended (halo) region (:1059-1109).  Here: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests), the same two partitions the reference
offers -- horizontal tiles (MPI_COMM_d) and cyclic dealing of points (MPI_COMM_e, common_mpi_scale.f90:1428-1455).
Plumbing only: no numerics live here.
"""
import math

import torch
import torch.distributed as dist


def tile_grid(world):
    """px x py process lattice, as square as possible (PRC_NUM_X x PRC_NUM_Y)."""
    px = int(math.sqrt(world))
    while world % px:
        px -= 1
    return world // px, px


def tile_partition(nx, ny, world):
    """Horizontal tiles [i0, i1) x [j0, j1) per rank, rank = ix + px*iy."""
    px, py = tile_grid(world)
    xs = [round(nx * i / px) for i in range(px + 1)]
    ys = [round(ny * j / py) for j in range(py + 1)]
    return [(xs[r % px], xs[r % px + 1], ys[r // px], ys[r // px + 1]) for r in range(world)]


def cyclic_points(nij, rank, world):
    """The reference's cyclic dealing: local point i of ensemble-rank m is subdomain point m + world*i
    (scale/common/common_mpi_scale.f90:1428-1455, grd_to_buf)."""
    return torch.arange(rank, nij, world, dtype=torch.int64)


def allgatherv_rows(shard, group=None):
    """ALLGATHERV of row blocks with different row counts per rank (letkf_obs.f90:1036-1046): all ranks end up
    with the rows of rank 0, then rank 1, ... .  Implemented as one all_gather of the counts and one
    all_gather_into_tensor on max-padded blocks (a single large collective per array: on xGMI a ring is per-link
    bound, so fewer, larger messages win), then a compaction on the receiving side."""
    world = dist.get_world_size(group)
    n = torch.tensor([shard.shape[0]], dtype=torch.int64, device=shard.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts) if counts else 0
    if nmax == 0:
        return shard.new_zeros((0,) + tuple(shard.shape[1:])), counts
    padded = shard.new_zeros((nmax,) + tuple(shard.shape[1:]))
    padded[: shard.shape[0]] = shard
    out = shard.new_empty((world * nmax,) + tuple(shard.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == nmax for c in counts):
        return out, counts
    keep = torch.cat([torch.arange(r * nmax, r * nmax + counts[r], device=shard.device) for r in range(world)])
    return out.index_select(0, keep), counts


def halo_rows(ri, rj, tile, halo_i, halo_j):
    """Indices of the gathered rows whose fractional grid position (ri, rj) lies in the tile extended by the
    localisation halo (the reference's extended subdomain, letkf_obs.f90:922-976, 1059-1109)."""
    i0, i1, j0, j1 = tile
    ok = (ri >= i0 - halo_i) & (ri < i1 + halo_i) & (rj >= j0 - halo_j) & (rj < j1 + halo_j)
    return ok.nonzero(as_tuple=False).squeeze(1)


def exchange_rows(send_blocks, recv_counts, group=None):
    """Pairwise exchange of row blocks: rank r hands send_blocks[q] to rank q and receives recv_counts[q] rows from it
    (its own block is copied).  The halo-only alternative to the ALLGATHERV above: every rank gets just the rows of its
    extended subdomain, from the ranks that own them -- on a fully connected xGMI node the pairs use different links at
    the same time.  One batch of point-to-point operations (ncclSend / ncclRecv grouped by torch on the GPU box)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    proto = send_blocks[rank]
    tail = tuple(proto.shape[1:])
    recv = [proto.new_empty((int(recv_counts[q]),) + tail) for q in range(world)]
    ops = []
    for q in range(world):
        if q == rank:
            recv[q].copy_(send_blocks[q])
            continue
        if send_blocks[q].shape[0] > 0:
            ops.append(dist.P2POp(dist.isend, send_blocks[q].contiguous(), q, group))
        if recv[q].shape[0] > 0:
            ops.append(dist.P2POp(dist.irecv, recv[q], q, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv


def nij1_of(nlon, nlat, rank, world):
    """points of a subdomain dealt to ensemble-rank `rank` (grd_to_buf: point ij goes to rank ij mod world)."""
    return (nlon * nlat - rank + world - 1) // world


def scatter_members_alltoall(ctx, nlev, nlon, nlat, nv3d, mstart, mcount, v3dg, x, nens, group=None):
    """scatter_grd_mpi_alltoall (scale/common/common_mpi_scale.f90:1279-1335): ranks 0 .. mcount-1 each hold ONE member's
    field v3dg(nlev,nlon,nlat,nv3d) (member mstart + rank; v3dg may be None on the others); afterwards every rank has,
    for ITS share of the grid points, the members mstart .. mstart+mcount-1 in slots of x = gues3d(nij1,nlev,nens,nv3d).
    The dealing of points into per-destination blocks and the filing into the member slot are the library's
    letkf_member_points_dev (the reference's grd_to_buf and its copy loop); the exchange is one batch of pairwise sends
    (MPI_ALLTOALL(V) of nij1max x nlevall blocks there; true counts here)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nij1 = nij1_of(nlon, nlat, rank, world)
    dev = x.device
    blocks = []
    for d in range(world):
        nd = nij1_of(nlon, nlat, d, world)
        if rank < mcount:
            b = torch.empty(nv3d * nlev * nd, dtype=x.dtype, device=dev)
            ctx.member_points(0, nlev, nlon, nlat, nv3d, world, d, 0, v3dg, b, nd, 1, 0, nd * nlev)
        else:
            b = torch.empty(0, dtype=x.dtype, device=dev)
        blocks.append(b)
    rc = [nv3d * nlev * nij1 if s < mcount else 0 for s in range(world)]
    got = exchange_rows(blocks, rc, group)
    xv = x.view(nv3d, nens, nlev * nij1)
    for s in range(mcount):
        xv[:, mstart + s, :] = got[s].view(nv3d, nlev * nij1)


def gather_members_alltoall(ctx, nlev, nlon, nlat, nv3d, mstart, mcount, x, nens, v3dg, group=None):
    """gather_grd_mpi_alltoall (:1340-1396), the way back: member mstart + r of every rank's x goes to rank r, which
    assembles the whole field v3dg (ranks >= mcount receive nothing)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nij1 = nij1_of(nlon, nlat, rank, world)
    xv = x.view(nv3d, nens, nlev * nij1)
    blocks = [(xv[:, mstart + d, :].contiguous().view(-1) if d < mcount else x.new_empty(0)) for d in range(world)]
    rc = [nv3d * nlev * nij1_of(nlon, nlat, s, world) if rank < mcount else 0 for s in range(world)]
    got = exchange_rows(blocks, rc, group)
    if rank < mcount:
        for s in range(world):
            ns = nij1_of(nlon, nlat, s, world)
            ctx.member_points(1, nlev, nlon, nlat, nv3d, world, s, 0, v3dg, got[s], ns, 1, 0, ns * nlev)
