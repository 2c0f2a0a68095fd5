#!/usr/bin/env python3
"""bench.py -- grid-point LETKF solves/s of the das_letkf-equivalent call on MI355X (BASELINE.json metric).

One "step" = one full analysis of the workload grid, through the C ABI with every input resident in HBM:
obs_local for all points on the device (letkf_obs_search_columns_dev: the local-observation lists are rebuilt inside
every timed step) followed by the batched loop body (letkf_das_points_dev: local-obs gather by index, k x k or n x n
eigen-solve, relaxation, transform of the nv = 11 variables).  At N = 1 the workload is BASELINE.json configs[1]
(240x240x60 grid, k = 50, ~200 local obs/point).  For N > 1 (torchrun, one rank per GPU, RCCL) every rank owns a
tile of that size (weak scaling) and each step starts with the path's one exchange: the all-gather of the
observation-table shards (the localisation-halo exchange of scale/letkf/letkf_obs.f90:1036-1046).

Prints ONE JSON line on rank 0: the contract fields, `roofline` for the dominant kernel (its name and average launch
duration come from the library: HIP events on the launch stream), `cpu_baseline` (the oracle's OpenMP restatement on a
bounded sample of the same points, kind "port") with `parity_sample_max_rel` = the GPU analysis against that very
oracle run (exit status 3 if it exceeds 1e-10), and `cpu_baseline_reference` (the reference's own letkf_core compiled
into oracle/_ref, kind "reference", when that library travelled with the repository).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--warm-runs", choices=["auto", "x", "z"], default="auto",
                    help="direction of the eigensolver's warm-start runs: along ij (letkf_das_args.warm_stride = 0) or up "
                         "the columns (warm_stride = nij1); auto = whichever neighbour is closer in localisation "
                         "scales (mean level spacing / vertical scale against dx / horizontal scale)")
    ap.add_argument("--warm-run", type=int, default=0, help="run length (letkf_das_args.warm_run; 0 = library default)")
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--relax", default="rtps", choices=["rtps", "rtpp", "none"])
    ap.add_argument("--list-gb", type=float, default=8.0, help="--lists pipeline: list workspace of the library in GiB")
    ap.add_argument("--lists", default="columns", choices=["columns", "search", "torch", "fused", "pipeline"],
                    help="where the local-obs lists come from: letkf_obs_search_columns_dev (default), the per-point "
                         "letkf_obs_search_dev, the torch workload builder (no search on the device), or obs_local "
                         "fused into the loop-body kernel; pipeline = letkf_das_columns_dev: ONE call per step, column search + loop "
                         "body by slabs of levels whose lists fit --list-gb of library workspace (no lists on the host side)")
    ap.add_argument("--no-search-in-step", action="store_true",
                    help="build the lists once, outside the timed region (times the loop body alone)")
    ap.add_argument("--ensval", default="iid", choices=["iid", "correlated"],
                    help="obs-space perturbations: independent draws, or an H-like combination of the (spatially "
                         "smooth) state perturbations around every observation + noise (SURVEY.md section 8(d))")
    ap.add_argument("--obs-spread", type=float, default=0.0,
                    help="scale the obs-space perturbations so that their standard deviation is this many observation "
                         "errors (0: as generated -- 0.67 for iid, 0.81 for correlated); 2-3 is ordinary for radar "
                         "reflectivity in convection")
    ap.add_argument("--level-slab", type=int, default=0,
                    help="analyse the domain L levels at a time (the reference's level loop, scale/letkf/letkf_tools.f90:313): "
                         "per step and slab obs_local (column search) for the slab's points, then the loop body -- the local-"
                         "observation lists exist for one slab only (C4-gpu: 10 M points x ~4900 x 20 B do not fit at once)")
    ap.add_argument("--no-torch-lists", action="store_true",
                    help="with --lists fused: do not build the local-observation lists in torch either (workloads whose lists do "
                         "not fit: the fused loop body never needs them; the mean list length is taken from its nobs_out)")
    ap.add_argument("--state-slab", action="store_true",
                    help="with --level-slab: the ensemble state of a slab is copied into a compact slab buffer before its loop body "
                         "and the analysis is written to a slab buffer (the state streamed by level: C5-gpu's first guess + analysis "
                         "are 2 x 152 GB; the copy stands for the transfer and is inside the timed step)")
    ap.add_argument("--max-nobs", type=int, default=0,
                    help="MAX_NOBS_PER_GRID: two radar ctypes on the lattice, each limited to this many observations")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank owns a domain of the workload's size (the contract line); strong = ONE "
                         "domain cut into tiles, each rank running set_letkf_obs (sort, exchange, halo plan) + obs_local + "
                         "the loop body for its tile (bench_tiles.py)")
    ap.add_argument("--state-layout", default="ref", choices=["ref", "member"],
                    help="ensemble state in HBM: the reference's gues3d(nij1*nlev, nens, nv3d) (point-fastest) or the "
                         "point-major member-fastest layout the ABI's strides also allow (sm = 1)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "lib", "halo"],
                    help="N > 1: the obs all-gather through torch.distributed (default) or through the library's own "
                         "letkf_obs_allgatherv_dev on an RCCL communicator this script creates (ncclCommInitRank); "
                         "halo (--scaling strong only): pairwise sends of just the rows each extended subdomain holds")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="N=V",
                    help="letkf_ctx_set_option(N, V) on the bench's context (include/letkf_amd.h LETKF_OPT_*), e.g. 2=1: the list-free "
                         "route of --lists pipeline wherever it is eligible")
    ap.add_argument("--eigen-stage-only", action="store_true",
                    help="measurement of the fallback: LETKF_OPT_STAGED_POLY = 0, every staged point through the eigen stage "
                         "(workgroup Jacobi at orders <= 208, block Jacobi above) instead of the eigen-free route")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: min(affinity, cgroup quota, 16 = the box's CPU share)")
    args = ap.parse_args()

    # --gpus N is the contract; WORLD_SIZE is how the ranks learn it.  Started bare with N > 1 (no torchrun), launch
    # the N ranks as a CHILD process -- before torch or HIP is touched in this one -- and hand back its exit code.
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and env_world == 1 and "RANK" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if env_world != max(args.gpus, 1):
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with "
                 f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")

    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    import bench_workload as bw

    world = env_world
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists)"
    # LETKF_BENCH_BACKEND=gloo: rehearsal of the N > 1 paths on a box with fewer GPUs than ranks (the ranks share the
    # devices round-robin and exchange through gloo; RCCL refuses two ranks on one device).  Not a measurement.
    backend = os.environ.get("LETKF_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_gpus = world

    pkg = load_package()
    if rank == 0:
        pkg.build()                 # one rank runs make; the others wait (no race on a stale .so)
    if world > 1:
        dist.barrier()
    stream = torch.cuda.current_stream()
    ctx = pkg.Context(local_rank, stream.cuda_stream)
    if args.eigen_stage_only:
        ctx.set_option(ctx.OPT_STAGED_POLY, 0)
    for ov in args.ctx_option:
        o_, v_ = ov.split("=")
        ctx.set_option(int(o_), int(v_))

    if args.scaling == "strong":
        if args.lists not in ("columns", "pipeline"):
            sys.exit("--scaling strong: --lists columns (search + loop body) or pipeline (letkf_das_columns_dev)")
        import bench_tiles
        r = bench_tiles.run(args, ctx, pkg, dev, rank, world)
        if rank == 0:
            k_, nv_ = r["k"], r["nv"]
            b_alg = bw.alg_bytes_per_solve(r["n_mean"], k_, nv_)
            f_alg = bw.alg_flops_required(r["n_mean"], k_, nv_, rtps=(args.relax == "rtps"))
            per_rank = r["npts_total"] / world
            ks = r["kern_ms"] * 1e-3
            tfl = f_alg * per_rank / ks / 1e12 if ks > 0 else None
            print(json.dumps({
                "metric": "grid-point LETKF solves/sec", "value": r["npts_total"] * args.steps / r["elapsed"],
                "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": r["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{args.workload}: ONE {bw.CONFIGS[args.workload]['nx']}x{bw.CONFIGS[args.workload]['ny']}x"
                                       f"{bw.CONFIGS[args.workload]['nz']} domain as {r['tiles']}, k={k_}, nv={nv_}, mean "
                                       f"{r['n_mean']:.1f} local obs/point, relax={args.relax}; per step and rank: mesh sort, "
                                       f"obs exchange ({args.exchange}), halo plan, "
                                       + ("obs_local + loop body in one call (letkf_das_columns_dev)" if args.lists == "pipeline" else "obs_local, loop body"),
                           "points_total": r["npts_total"], "obs_rows_global": r["nobs"],
                           "obs_rows_per_rank_with_halo": r["halo_rows_mean"],
                           "obs_rows_received_per_rank": r["rows_received_mean"], "parallelism": f"tiles x{world}"},
                "anal_checksum": r["anal_checksum"],
                "nonzero_status_points": r["bad"], "jacobi_sweeps_mean": r["sweeps_mean"],
                "roofline": {"bound": "mfma", "achieved": tfl, "peak": 78.6, "unit": "TFLOP/s",
                             "frac": (tfl / 78.6) if tfl else None, "traffic": None, "kernel": ctx.last_path(),
                             "kernel_ms": r["kern_ms"], "launches": r["nlaunch"], "alg_bytes_per_solve": b_alg,
                             "alg_flops_per_solve": f_alg, "note": "loop-body launch of the slowest rank, per GPU"},
                "cpu_baseline": None}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return None
    if args.max_nobs > 0 and args.lists not in ("columns", "search", "pipeline"):
        sys.exit("--max-nobs needs --lists columns, search or pipeline (the device obs_local)")
    slab_mode = args.level_slab > 0
    if slab_mode and (args.lists != "columns" or args.no_search_in_step or world > 1 or args.state_layout != "ref"):
        sys.exit("--level-slab: the column search inside the step, one GPU, the reference's state layout")
    if args.no_torch_lists and args.lists not in ("fused", "pipeline"):
        sys.exit("--no-torch-lists goes with --lists fused / pipeline")
    w = bw.build(args.workload, dev, rank=rank, world=world, ensval_kind=args.ensval, lists=not (slab_mode or args.no_torch_lists))
    k, nv, npts = w["k"], w["nv"], w["npts"]
    bw.relayout_state(w, args.state_layout)
    # the streaming passes either side of the loop: mean into slot k, members -> perturbations
    ctx.ens_mean(k, nv, npts, w["gues"], w["sp"], w["sm"], w["sv"])
    ctx.to_perturbations(k, nv, npts, w["gues"], w["sp"], w["sm"], w["sv"])
    if args.ensval == "correlated":
        bw.correlate_ensval(w)      # needs the perturbations
    spread_built = float(w["ensval"][:, :k].std().item()) / w["cfg"]["err"]
    if args.obs_spread > 0.0:
        w["ensval"][:, :k] *= args.obs_spread / spread_built
        w["dep"] *= ((1.0 + args.obs_spread ** 2) / (1.0 + spread_built ** 2)) ** 0.5   # departures ~ N(0, err^2 + spread^2)
    obs_spread = args.obs_spread if args.obs_spread > 0.0 else spread_built
    search_ms = None
    host_lists_ok = True
    in_step = args.lists in ("search", "columns") and not args.no_search_in_step
    if args.lists != "torch":
        # the table as set_letkf_obs leaves it behind: rows in mesh order, prefix sums per cell
        t_s, keep_s, order_s, pts_s = bw.search_tables(w, pkg, dev, max_nobs=args.max_nobs)
        n_rows_unsorted = int(w["ensval"].shape[0])
        w["ensval"] = w["ensval"][order_s].contiguous()
        w["dep"] = w["dep"][order_s].contiguous()
        if args.lists in ("fused", "pipeline") and w.get("obs_idx") is not None:
            # the torch-built lists (kept for the CPU checker: these two modes have no lists of their own on the host side) index
            # the table in its ORIGINAL order: through the inverse of the sort (the table may have lost rows on the way: a list
            # that names one cannot be checked)
            w["obs_idx"], host_lists_ok = bw.remap_lists_to_sorted(w["obs_idx"], order_s, n_rows_unsorted)
    if args.lists in ("search", "columns") and slab_mode:
        nij_s = w["cfg"]["nx"] * w["cfg"]["ny"]
        rig_s, rjg_s = pts_s[0][:nij_s].contiguous(), pts_s[1][:nij_s].contiguous()
    elif args.lists in ("search", "columns"):
        n_torch = int(w["obs_off"][-1].item())
        nij_s = w["cfg"]["nx"] * w["cfg"]["ny"]
        rig_s, rjg_s = pts_s[0][:nij_s].contiguous(), pts_s[1][:nij_s].contiguous()

        def do_search():
            if args.lists == "columns":   # one wave per horizontal point, all levels (letkf_obs_search_columns_dev)
                return ctx.obs_search_columns(t_s, nij_s, w["cfg"]["nz"], rig_s, rjg_s, pts_s[2], pts_s[3])
            return ctx.obs_search(t_s, *pts_s)

        for rep in range(3):   # (the last one runs on blocks the caching allocator already owns)
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            off_s, idx_s, rd_s, rl_s = do_search()
            torch.cuda.synchronize()
            search_ms = (time.perf_counter() - t0s) * 1e3
        if args.max_nobs == 0:
            assert int(off_s[-1].item()) == n_torch, "device search and torch builder disagree on the list sizes"
        w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"] = off_s, idx_s, rd_s, rl_s
        cnt = (off_s[1:] - off_s[:-1])
        w["n_mean"], w["n_max"] = float(cnt.double().mean()), int(cnt.max())
    nens = w["nens"]
    slabs, slab_stat = [], {"nnz": 0, "n_max": 0, "last": None}
    if slab_mode:
        nz_ = w["cfg"]["nz"]
        slabs = [(l0, min(l0 + args.level_slab, nz_)) for l0 in range(0, nz_, args.level_slab)]
        npmax = nij_s * args.level_slab
        infl_slab = torch.ones(npmax * nv, dtype=torch.float64, device=dev)
        if args.state_slab:
            gbuf = torch.empty(nv * nens * npmax, dtype=torch.float64, device=dev)
            abuf = torch.empty_like(gbuf)
    anal = torch.empty_like(w["gues"]) if not (slab_mode and args.state_slab) else None
    infl = torch.ones(npts * nv, dtype=torch.float64, device=dev) if not slab_mode else None
    status = torch.zeros(npts, dtype=torch.int32, device=dev)
    nsweep = torch.zeros(npts, dtype=torch.int32, device=dev)
    nobs_fused = torch.zeros(npts, dtype=torch.int32, device=dev) if args.lists in ("fused", "pipeline") else None
    if args.lists == "pipeline":
        nij_s = w["cfg"]["nx"] * w["cfg"]["ny"]
        rig_s, rjg_s = pts_s[0][:nij_s].contiguous(), pts_s[1][:nij_s].contiguous()
    relax = dict(rtps=dict(relax_alpha_spread=0.95), rtpp=dict(relax_alpha=0.7), none=dict())[args.relax]
    nij1 = w["cfg"]["nx"] * w["cfg"]["ny"]                    # points are p = ij + nij1 * lev (gues3d's order)
    zdir = args.warm_runs == "z"
    if args.warm_runs == "auto":
        c_ = w["cfg"]
        zl = bw.level_heights(c_["nz"], c_["ztop"])
        zdir = c_["nz"] > 1 and float(np.mean(np.diff(zl))) / c_["vloc"] < c_["dx"] / c_["hloc"]
    warm = dict(warm_run=args.warm_run, warm_stride=nij1 if (zdir and npts % nij1 == 0) else 0)

    # N>1: each rank owns a ragged 1/N shard of the rows of its obs table; every step starts with the path's one
    # exchange, the ALLGATHERV of those shards over RCCL (scale-letkf_amd/sharding.py, covered by the gloo test)
    shard = None
    if world > 1:
        import importlib
        sharding = importlib.import_module("scale_letkf_amd.sharding")
        rows = w["ensval"].shape[0]
        cuts = [round(rows * r / world) for r in range(world + 1)]
        shard = w["ensval"][cuts[rank]:cuts[rank + 1]].clone()
        if args.exchange == "lib":
            # the host's side of C-ABI section 8: an RCCL communicator of its own (the unique id travels through the
            # process group that is already up), handed to the library as a plain pointer
            import ctypes as C

            class UniqueId(C.Structure):
                _fields_ = [("internal", C.c_char * 128)]
            rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
            uid = UniqueId()
            if rank == 0:
                assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
            ub = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=dev)
            dist.broadcast(ub, 0)
            C.memmove(C.byref(uid), bytes(ub.cpu().tolist()), 128)
            ncomm = C.c_void_p()
            rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            assert rccl.ncclCommInitRank(C.byref(ncomm), world, uid, rank) == 0
            counts_x = [cuts[r + 1] - cuts[r] for r in range(world)]
            gathered = torch.empty_like(w["ensval"])

    def step_slabs():
        # the reference's level loop: per slab of levels obs_local for its points, then the loop body
        slab_stat["nnz"], slab_stat["n_max"] = 0, 0
        for (l0, l1) in slabs:
            p0, p1 = l0 * nij_s, l1 * nij_s
            np_ = p1 - p0
            off_, idx_, rd_, rl_ = ctx.obs_search_columns(t_s, nij_s, l1 - l0, rig_s, rjg_s, pts_s[2][p0:p1], pts_s[3][p0:p1])
            slab_stat["nnz"] += int(idx_.numel())
            ws_ = dict(warm_run=args.warm_run, warm_stride=nij_s if (warm["warm_stride"] and l1 - l0 > 1) else 0)
            if args.state_slab:
                gs, as_ = gbuf[: nv * nens * np_], abuf[: nv * nens * np_]
                gs.view(nv, nens, np_).copy_(bw.state_view(w, w["gues"])[:, :, p0:p1])     # the slab's state "arrives"
                ssp, ssm, ssv = 1, np_, np_ * nens
            else:
                gs, as_ = w["gues"][p0:], anal[p0:]
                ssp, ssm, ssv = w["sp"], w["sm"], w["sv"]
            ctx.das_points(k, nv, off_, idx_, rd_, rl_, w["ensval"], w["kld"], w["dep"], infl_slab[: np_ * nv], gs, as_,
                           ssp, ssm, ssv, status=status[p0:p1], nsweep=nsweep[p0:p1], **ws_, **relax)
            slab_stat["last"] = (p0, p1, off_, idx_, rd_, rl_, gs, as_, ssp, ssm, ssv)

    def step():
        if slab_mode:
            return step_slabs()
        ens = w["ensval"]
        if world > 1:
            if args.exchange == "lib":
                ctx.obs_allgatherv(ncomm.value, rank, counts_x, shard, gathered)
                ens = gathered
            else:
                ens, _ = sharding.allgatherv_rows(shard)
        if in_step:
            # the whole das_letkf-equivalent call: obs_local for every point, then the batched loop body
            w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"] = do_search()
        if args.lists == "pipeline":
            ctx.das_columns(k, nv, t_s, nij_s, w["cfg"]["nz"], rig_s, rjg_s, pts_s[2], pts_s[3], ens, w["kld"], w["dep"], infl,
                            w["gues"], anal, w["sp"], w["sm"], w["sv"], list_bytes=int(args.list_gb * 2 ** 30), nobs_out=nobs_fused,
                            status=status, nsweep=nsweep, warm_run=args.warm_run, **relax)
        elif args.lists == "fused":
            ctx.das_points(k, nv, None, None, None, None, ens, w["kld"], w["dep"], infl, w["gues"], anal, w["sp"],
                           w["sm"], w["sv"], status=status, nsweep=nsweep, fused=(t_s, *pts_s), nobs_out=nobs_fused, **warm, **relax)
        else:
            ctx.das_points(k, nv, w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"], ens, w["kld"], w["dep"], infl,
                           w["gues"], anal, w["sp"], w["sm"], w["sv"], status=status, nsweep=nsweep, **warm, **relax)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(True)
    ctx.timing_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    kern_ms, nlaunch = ctx.timing_read(reset=True)
    ctx.timing_enable(False)
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    bad = int((status != 0).sum().item())
    sweeps_mean = float(nsweep.clamp(min=0).double().mean().item())
    # points of the staged path analysed without an eigen stage report -(CG iterations) (include/letkf_amd.h)
    npoly = int((nsweep < 0).sum().item())
    cheb_deg_mean = float((-nsweep[nsweep < 0]).double().mean().item()) if npoly else None
    cheb_deg_max = int((-nsweep[nsweep < 0]).max().item()) if npoly else None
    solves = npts * args.steps * world
    value = solves / elapsed

    out = None
    rc = 0
    if args.lists in ("fused", "pipeline"):
        w["n_mean"], w["n_max"] = float(nobs_fused.double().mean().item()), int(nobs_fused.max().item())
    if slab_mode:
        w["n_mean"] = slab_stat["nnz"] / npts
        p0, p1, off_, idx_, rd_, rl_, gs, as_, ssp, ssm, ssv = slab_stat["last"]
        cnt_ = off_[1:] - off_[:-1]
        w["n_max"] = int(cnt_.max())
        # the CPU baseline / parity sample is drawn from the last slab (the only lists and -- with --state-slab -- the only
        # analysis that exist at the end of a step)
        w_chk = dict(w, npts=p1 - p0, obs_off=off_, obs_idx=idx_, rdiag=rd_, rloc=rl_, gues=gs, sp=ssp, sm=ssm, sv=ssv,
                     n_mean=float(cnt_.double().mean()))
        anal_chk = as_
    else:
        w_chk, anal_chk = w, anal
    if rank == 0:
        n_mean = w["n_mean"]
        b_alg = bw.alg_bytes_per_solve(n_mean, k, nv)
        # algorithmic flops: SURVEY 8(d)'s k x k count, or -- for a workload with fewer local observations than members --
        # the count of the n x n formulation when that is the smaller one (what the analysis requires, not what a
        # k x k solver would spend); the k x k figure rides along as alg_flops_kxk_nominal
        f_kxk = bw.alg_flops_per_solve(n_mean, k, nv, rtps=(args.relax == "rtps"))
        f_alg = bw.alg_flops_required(n_mean, k, nv, rtps=(args.relax == "rtps"))
        if cheb_deg_mean is not None and npoly == npts:
            # the eigen-free formulation needs fewer flops still (no 9 n^3): price the kernel with what it executes
            f_alg = min(f_alg, bw.alg_flops_poly(n_mean, k, nv, cheb_deg_mean))
        # (the loop body may be several launches per step: one per level slab)
        kern_launch_ms = kern_ms
        kern_ms = kern_ms * nlaunch / max(args.steps, 1)
        kern_s = kern_ms * 1e-3
        achieved = b_alg * npts / kern_s / 1e9 if kern_s > 0 else None
        traffic = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_member.json" if args.state_layout == "member" else "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if (tj.get("workload") == args.workload and args.ensval == "iid" and args.max_nobs == 0
                        and tj.get("state_layout", "ref") == args.state_layout):
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = (f"{os.path.relpath(tpath, ROOT)}: rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, "
                                      f"MI355X_MICROARCH.md) of build {tj.get('build', '?')} on this workload -- a recorded constant, "
                                      f"NOT measured by this run")
            except Exception:
                traffic = None
        # Which roof bounds the kernel: arithmetic intensity of the algorithmic work against the ridge of the chip
        # (78.6 Tflop/s FP64 -- vector and matrix-core peak are the same figure -- over 8 TB/s = 9.8 flop/B).
        tflops = (f_alg * npts / kern_s / 1e12) if kern_s > 0 else None
        hbm = {"achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": (achieved / 8000.0) if achieved else None}
        fp64 = {"achieved": tflops, "peak": 78.6, "unit": "TFLOP/s", "frac": (tflops / 78.6) if tflops else None}
        compute_bound = f_alg / b_alg > 78.6e12 / 8.0e12
        main_ = fp64 if compute_bound else hbm
        roofline = {"bound": "mfma" if compute_bound else "hbm", "achieved": main_["achieved"], "peak": main_["peak"],
                    "unit": main_["unit"], "frac": main_["frac"], "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": ctx.last_path(), "kernel_ms": kern_ms, "kernel_ms_per_launch": kern_launch_ms,
                    "launches": nlaunch, "alg_bytes_per_solve": b_alg, "alg_flops_per_solve": f_alg,
                    "alg_flops_kxk_nominal": f_kxk,
                    "arithmetic_intensity": f_alg / b_alg, "fp64": fp64, "hbm": hbm,
                    "note": "kernel_ms = loop-body time per step (all its launches: one per level slab); bound = the roof the algorithmic intensity puts the kernel under; 'mfma' stands for the "
                            "FP64 peak (78.6 TFLOP/s, same for v_fma_f64 and v_mfma_f64): the Gram runs on the matrix "
                            "cores, the eigensolve on the vector ALUs; kernel_ms = the loop-body launch(es) only "
                            "(HIP events), ms_per_step also holds obs_local when search_in_step"}
        cpu = cpu_ref = None
        parity = None
        parity_note = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            # The checker's local-observation lists come from the ORACLE's obs_local (oracle/letkf_oracle.c orc_obs_local, limit
            # and criterion included) on a host copy of the same tables -- not from the device search, whichever route the step
            # took: the list-free and pipeline routes and the runs under MAX_NOBS_PER_GRID are checked like every other.
            # (--lists torch has no tables: the torch builder's lists serve.)
            chk = None
            if args.lists != "torch":
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import _search
                h_, alive_ = _search.host_struct_from_torch(t_s, keep_s)
                lo_ = slab_stat["last"][0] if slab_mode else 0
                hi_ = slab_stat["last"][1] if slab_mode else npts
                chk = dict(tables=h_, alive=alive_, coords=[c_[lo_:hi_].cpu().numpy() for c_ in pts_s], limited=args.max_nobs > 0)
            if chk is not None or (w_chk.get("obs_off") is not None and host_lists_ok):
                cpu, parity, parity_note = cpu_baseline(w_chk, relax, args.cpu_seconds, args.cpu_threads, anal_chk, chk)
                if w_chk.get("obs_off") is not None and host_lists_ok:
                    cpu_ref = cpu_baseline_reference(w_chk, args.cpu_threads, min(args.cpu_seconds, 10.0))
                if parity is not None and not (parity <= 1e-10):
                    rc = 3
        out = {"metric": "grid-point LETKF solves/sec", "value": value, "unit": "solves/s", "n_gpus": n_gpus,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": f"{args.workload}: {w['cfg']['nx']}x{w['cfg']['ny']}x{w['cfg']['nz']} grid, "
                                      f"k={k} members, nv={nv}, mean {n_mean:.1f} (max {w['n_max']}) local obs/point, "
                                      f"relax={args.relax}, ensval={args.ensval}, obs-space spread {obs_spread:.2f} obs errors" + (", state member-fastest" if args.state_layout == "member" else "")
                                      + (f", MAX_NOBS_PER_GRID={args.max_nobs} x 2 ctypes" if args.max_nobs else "")
                                      + (", eigen stage only (LETKF_OPT_STAGED_POLY = 0)" if args.eigen_stage_only else "")
                                      + (f"; analysed in slabs of {args.level_slab} level(s): obs_local + loop body per slab"
                                         + (", state streamed per slab" if args.state_slab else "") if slab_mode else ""),
                          "points_per_gpu": npts, "obs_table_rows": int(w["ensval"].shape[0]),
                          "parallelism": f"grid-point shard x{n_gpus}" + ((" + RCCL obs all-gather (" + args.exchange + ")") if world > 1 else "")},
               "warm_runs": ("columns (warm_stride = nij1)" if warm["warm_stride"] else "along ij"),
               "analysis_wall_s": elapsed / args.steps, "cycle_ms": elapsed / args.steps * 1e3,
               "solve_only_solves_per_s": (npts * world / kern_s) if kern_s > 0 else None,
               "nonzero_status_points": bad,
               "jacobi_sweeps_mean": sweeps_mean, "eigenfree_points": npoly, "eigenfree_iterations_mean": cheb_deg_mean, "eigenfree_iterations_max": cheb_deg_max,
               "eigenfree_fallback_points": (int(((nsweep > 0) & (status == 0)).sum().item()) if npoly else 0),
               "obs_spread": obs_spread,
               "lists": args.lists, "search_ms": search_ms,
               "search_in_step": bool(in_step),
               "parity_sample_max_rel": parity, "parity_tolerance": 1e-10, "parity_lists": parity_note,
               "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_reference": cpu_ref}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)
    return out


def host_threads(requested):
    if requested > 0:
        return requested
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return min(n, 16)   # a one-GPU box's CPU share is 16 cores


def cpu_baseline(w, relax, seconds, threads, anal, chk=None):
    """The oracle's OpenMP restatement of the same loop body on a bounded sample of the same points (kind "port"),
    timed on this box's host cores -- and the GPU analysis of those very points checked against it
    (max over variables of |d xa| / max(|x-bar|, |x'|)).  A reported baseline, not the target.
    chk = dict(tables, coords, limited): the sample's local-observation lists are built by the oracle's own obs_local
    (orc_obs_local: scale/letkf/letkf_tools.f90:1325-1759) on the host copy `tables` of the search tables, for the points at
    `coords` -- the check is then independent of the device search.  Under MAX_NOBS_PER_GRID a point whose selection falls
    between EQUAL keys at the threshold is left out (the reference's quick-select is unstable there, common_sort.f90:341-369:
    either choice is its result, and a lattice has such points).  The list building is not part of the timed baseline."""
    import numpy as np
    import torch
    import bench_workload as bw
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    ncores = host_threads(threads)
    k, nv, npts, nens = w["k"], w["nv"], w["npts"], w["nens"]
    ens = w["ensval"].cpu().numpy()
    dep = w["dep"].cpu().numpy()
    rng = np.random.default_rng(1)
    note = {"lists": "device search / torch builder", "points": 0, "left_out_for_ties": 0}

    def sample(pts):
        if chk is None:
            return bw.sample_points(w, pts)
        import _search
        co = chk["coords"]
        off, idx, rd, rl, tied = _search.oracle_csr(chk["tables"], co[0][pts], co[1][pts], co[2][pts], co[3][pts], nthreads=ncores)
        if chk["limited"] and tied.any():
            keep = tied == 0
            ent = np.repeat(keep, np.diff(off))
            cnt = np.diff(off)[keep]
            off = np.zeros(len(cnt) + 1, dtype=np.int64)
            off[1:] = np.cumsum(cnt)
            idx, rd, rl, pts = idx[ent], rd[ent], rl[ent], pts[keep]
            note["left_out_for_ties"] += int((~keep).sum())
        note["lists"] = "oracle obs_local (orc_obs_local) on a host copy of the tables"
        tp = torch.from_numpy(pts).to(w["gues"].device)
        gv = bw.state_view(w, w["gues"])[:, :, tp].contiguous().cpu().numpy().reshape(-1)
        return dict(off=off, idx=idx, rdiag=rd, rloc=rl, gues=gv, ns=len(pts), pts=pts)

    def run(ns):
        s = sample(np.sort(rng.choice(npts, size=ns, replace=False)))
        pts, ns = s["pts"], s["ns"]
        note["points"] += ns
        prm = _oracle.DasParams(k=k, nv=nv, det_run=0, infl_adaptive=0, relax_to_inflated_prior=0,
                                relax_alpha=relax.get("relax_alpha", 0.0),
                                relax_alpha_spread=relax.get("relax_alpha_spread", 0.0), q_update_top=0.0,
                                q_sprd_max=0.0, iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=ncores)
        t0 = time.perf_counter()
        r = _oracle.das_points(prm, s["off"], s["idx"], s["rdiag"], s["rloc"], ens, dep, None, np.ones(ns * nv),
                               s["gues"], 1, ns, ns * nens)
        dt = time.perf_counter() - t0
        assert r["rc"] == 0
        # parity of the GPU result on the same points
        tp = torch.from_numpy(pts).to(anal.device)
        got = bw.state_view(w, anal)[:, :k, tp].cpu().numpy()
        exp = r["anal"].reshape(nv, nens, ns)[:, :k]
        x = s["gues"].reshape(nv, nens, ns)
        worst = 0.0
        for v in range(nv):
            scale = max(np.abs(x[v, k]).max(), np.abs(x[v, :k]).max())
            worst = max(worst, float(np.abs(got[v] - exp[v]).max() / scale))
        return ns / dt, worst

    first = (64 if k <= 100 else 8 if k <= 400 else 2) * ncores
    rate0, p0 = run(min(npts, first))
    ns = int(min(npts, max(first, rate0 * seconds)))
    rate, p1 = run(ns)
    return ({"value": rate, "unit": "solves/s", "cores": ncores, "kind": "port",
             "sample": f"{ns} randomly chosen grid points of the same workload (all {nv} variables, same relaxation), "
                       f"oracle/letkf_oracle.c orc_das_letkf_points, OpenMP dynamic over points"}, max(p0, p1), note)


def cpu_baseline_reference(w, threads, seconds):
    """The reference's own letkf_core (common/common_letkf.f90 compiled into oracle/_ref/libletkf_ref.so, EISPACK rs +
    the reference DGEMM) on local-observation slices of sampled points of this workload: kind "reference".  It times
    letkf_core alone (the loop body's relaxation and transform are not part of the compiled reference); problems are
    dealt to `cores` host threads, each running ref_letkf_core_loop on its share.  None when oracle/_ref is absent."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    lib = _oracle.ref()
    if lib is None:
        return None
    ncores = host_threads(threads)
    k, npts = w["k"], w["npts"]
    off = w["obs_off"]
    rng = np.random.default_rng(2)
    n_fix = max(1, int(round(w["n_mean"])))
    threading.stack_size(512 * 1024 * 1024 if k > 200 else 64 * 1024 * 1024)   # letkf_core's automatic arrays

    def make(nprob):
        cand = rng.choice(npts, size=min(npts, 4 * nprob), replace=False)
        cnt = (off[cand + 1] - off[cand]).cpu().numpy() if hasattr(off, "cpu") else off[cand + 1] - off[cand]
        cand = cand[cnt >= n_fix][:nprob]
        if len(cand) == 0:
            return None
        nprob = len(cand)
        H = np.empty((nprob, k, n_fix))
        rd = np.empty((nprob, n_fix))
        rl = np.empty((nprob, n_fix))
        dp = np.empty((nprob, n_fix))
        ens = w["ensval"]
        for i, p in enumerate(cand):
            o0 = int(off[p])
            idx = w["obs_idx"][o0:o0 + n_fix].long()
            H[i] = ens[idx, :k].T.cpu().numpy()
            rd[i] = w["rdiag"][o0:o0 + n_fix].cpu().numpy()
            rl[i] = w["rloc"][o0:o0 + n_fix].cpu().numpy()
            dp[i] = w["dep"][idx].cpu().numpy()
        return H, rd, rl, dp

    def run(nprob):
        m = make(nprob)
        if m is None:
            return None, 0
        H, rd, rl, dp = m
        nprob = H.shape[0]
        infl = np.ones(nprob)
        trans = np.empty((nprob, k, k))
        transm = np.empty((nprob, k))
        pao = np.empty((nprob, k, k))
        dpt = C.POINTER(C.c_double)
        f = lambda a, i0: (a[i0:].ctypes.data_as(dpt))
        cuts = [round(nprob * r / ncores) for r in range(ncores + 1)]

        def work(r):
            i0, i1 = cuts[r], cuts[r + 1]
            if i1 > i0:
                lib.ref_letkf_core_loop(C.c_int(i1 - i0), C.c_int(k), C.c_int(n_fix), C.c_int(n_fix), f(H, i0), f(rd, i0),
                                        f(rl, i0), f(dp, i0), f(infl, i0), f(trans, i0), f(transm, i0), f(pao, i0))
        th = [threading.Thread(target=work, args=(r,)) for r in range(ncores)]
        t0 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        return nprob / (time.perf_counter() - t0), nprob

    rate0, n0 = run(4 * ncores)
    if rate0 is None:
        return None
    nprob = int(max(4 * ncores, min(20000, rate0 * seconds)))
    rate, n1 = run(nprob)
    return {"value": rate, "unit": "letkf_core calls/s", "cores": ncores, "kind": "reference",
            "sample": f"{n1} letkf_core problems (k={k}, the first {n_fix} local observations of sampled points of this "
                      f"workload, transm + pao returned, rdiag_wloc) through ref_letkf_core_loop of oracle/_ref/"
                      f"libletkf_ref.so = /root/reference/common/common_letkf.f90 + common_mtx.f90 + netlib.f + "
                      f"netlibblas.f compiled with amdflang -O2; {ncores} host threads, one share of the problems each; "
                      f"letkf_core only (no relaxation / transform)"}


if __name__ == "__main__":
    main()
