#!/usr/bin/env python3
"""bench.py -- grid-point LETKF solves/s of the batched das_letkf body on MI355X (BASELINE.json metric).

One "step" = one full analysis of the workload grid (every (ij, ilev) point: local-obs gather by index +
k x k eigen-solve + relaxation + transform of the nv=11 variables), through the C ABI
(letkf_das_points_dev), inputs resident in HBM.  At N=1 the workload is BASELINE.json configs[1]
(240x240x60 grid, k=50, ~200 local obs/point).  For N>1 (torchrun, one rank per GPU, RCCL) every rank owns a
tile of that size (weak scaling) and each step starts with the path's one exchange: the all-gather of the
observation table shards (the localisation-halo exchange of scale/letkf/letkf_obs.f90:1036-1046).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--relax", default="rtps", choices=["rtps", "rtpp", "none"])
    ap.add_argument("--lists", default="torch", choices=["torch", "search", "columns", "fused"],
                    help="where the local-obs lists come from: the torch workload builder, or letkf_obs_search_dev "
                         "(on-device obs_local; its time is reported separately as search_ms)")
    ap.add_argument("--search-in-step", action="store_true",
                    help="with --lists search: rebuild the local lists with the device search inside every timed step")
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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    n_gpus = world

    pkg = load_package()
    if rank == 0:
        pkg.build()                 # one rank runs make; the others wait (no race on a stale .so)
    if world > 1:
        dist.barrier()
    stream = torch.cuda.current_stream()
    ctx = pkg.Context(local_rank, stream.cuda_stream)

    w = bw.build(args.workload, dev, rank=rank, world=world)
    k, nv, npts = w["k"], w["nv"], w["npts"]
    # the streaming passes either side of the loop: mean into slot k, members -> perturbations
    ctx.ens_mean(k, nv, npts, w["gues"], w["sp"], w["sm"], w["sv"])
    ctx.to_perturbations(k, nv, npts, w["gues"], w["sp"], w["sm"], w["sv"])
    search_ms = None
    if args.lists == "fused":
        # obs_local fused into the loop body (letkf_das_points_fused_dev): no lists at all
        t_s, keep_s, order_s, pts_s = bw.search_tables(w, pkg, dev)
        w["ensval"] = w["ensval"][order_s].contiguous()
        w["dep"] = w["dep"][order_s].contiguous()
    if args.lists in ("search", "columns"):
        # obs_local on the device: rebuild the lists with the search kernel on the mesh-sorted table
        t_s, keep_s, order_s, pts_s = bw.search_tables(w, pkg, dev)
        n_torch = int(w["obs_off"][-1].item())
        w["ensval"] = w["ensval"][order_s].contiguous()
        w["dep"] = w["dep"][order_s].contiguous()
        nij_s = w["cfg"]["nx"] * w["cfg"]["ny"]

        def do_search():
            if args.lists == "columns":   # one wave per horizontal point, all levels (letkf_obs_search_columns_dev)
                return ctx.obs_search_columns(t_s, nij_s, w["cfg"]["nz"], pts_s[0][:nij_s].contiguous(),
                                              pts_s[1][:nij_s].contiguous(), pts_s[2], pts_s[3])
            return ctx.obs_search(t_s, *pts_s)

        for rep in range(2):
            torch.cuda.synchronize()
            t0s = time.perf_counter()
            off_s, idx_s, rd_s, rl_s = do_search()
            torch.cuda.synchronize()
            search_ms = (time.perf_counter() - t0s) * 1e3
        assert int(off_s[-1].item()) == n_torch, "device search and torch builder disagree on the list sizes"
        w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"] = off_s, idx_s, rd_s, rl_s
    anal = torch.empty_like(w["gues"])
    infl = torch.ones(npts * nv, dtype=torch.float64, device=dev)
    status = torch.zeros(npts, dtype=torch.int32, device=dev)
    nsweep = torch.zeros(npts, dtype=torch.int32, device=dev)
    relax = dict(rtps=dict(relax_alpha_spread=0.95), rtpp=dict(relax_alpha=0.7), none=dict())[args.relax]

    # N>1: each rank owns a ragged 1/N shard of the rows of its obs table; every step starts with the path's one
    # exchange, the ALLGATHERV of those shards over RCCL (scale-letkf_amd/sharding.py, covered by the gloo test)
    shard = None
    if world > 1:
        import importlib
        sharding = importlib.import_module("scale_letkf_amd.sharding")
        rows = w["ensval"].shape[0]
        cuts = [round(rows * r / world) for r in range(world + 1)]
        shard = w["ensval"][cuts[rank]:cuts[rank + 1]].clone()

    def step():
        ens = w["ensval"]
        if world > 1:
            ens, _ = sharding.allgatherv_rows(shard)
        if args.lists in ("search", "columns") and args.search_in_step:
            # the whole das_letkf-equivalent call: obs_local for every point, then the batched loop body
            w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"] = do_search()
        if args.lists == "fused":
            ctx.das_points(k, nv, None, None, None, None, ens, w["kld"], w["dep"], infl, w["gues"], anal, w["sp"],
                           w["sm"], w["sv"], status=status, nsweep=nsweep, fused=(t_s, *pts_s), **relax)
        else:
            ctx.das_points(k, nv, w["obs_off"], w["obs_idx"], w["rdiag"], w["rloc"], ens, w["kld"], w["dep"], infl,
                           w["gues"], anal, w["sp"], w["sm"], w["sv"], status=status, nsweep=nsweep, **relax)

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
    sweeps_mean = float(nsweep.double().mean().item())
    solves = npts * args.steps * world
    value = solves / elapsed

    out = None
    if rank == 0:
        n_mean = w["n_mean"]
        b_alg = bw.alg_bytes_per_solve(n_mean, k, nv)
        f_alg = bw.alg_flops_per_solve(n_mean, k, nv, rtps=(args.relax == "rtps"))
        kern_s = kern_ms * 1e-3
        achieved = b_alg * npts / kern_s / 1e9 if kern_s > 0 else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # Which roof bounds the kernel: arithmetic intensity of the algorithmic work against the ridge of the chip
        # (78.6 Tflop/s FP64 -- vector and matrix-core peak are the same figure -- over 8 TB/s = 9.8 flop/B).
        tflops = (f_alg * npts / kern_s / 1e12) if kern_s > 0 else None
        hbm = {"achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": (achieved / 8000.0) if achieved else None}
        fp64 = {"achieved": tflops, "peak": 78.6, "unit": "TFLOP/s", "frac": (tflops / 78.6) if tflops else None}
        compute_bound = f_alg / b_alg > 78.6e12 / 8.0e12
        main = fp64 if compute_bound else hbm
        roofline = {"bound": "mfma" if compute_bound else "hbm", "achieved": main["achieved"], "peak": main["peak"],
                    "unit": main["unit"], "frac": main["frac"], "traffic": traffic,
                    "kernel": "letkf_wave_kernel<50,11>" if k <= 64 else "letkf_point_kernel", "kernel_ms": kern_ms,
                    "launches": nlaunch, "alg_bytes_per_solve": b_alg, "alg_flops_per_solve": f_alg,
                    "arithmetic_intensity": f_alg / b_alg, "fp64": fp64, "hbm": hbm,
                    "note": "bound = the roof the algorithmic intensity puts the kernel under; 'mfma' stands for the "
                            "FP64 peak (78.6 TFLOP/s, same for v_fma_f64 and v_mfma_f64): the Gram runs on the matrix "
                            "cores, the eigensolve on the vector ALUs"}
        cpu = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(w, relax, args.cpu_seconds, args.cpu_threads)
        out = {"metric": "grid-point LETKF solves/sec", "value": value, "unit": "solves/s", "n_gpus": n_gpus,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": f"{args.workload}: {w['cfg']['nx']}x{w['cfg']['ny']}x{w['cfg']['nz']} grid, "
                                      f"k={k} members, nv={nv}, mean {n_mean:.1f} (max {w['n_max']}) local obs/point, "
                                      f"relax={args.relax}", "points_per_gpu": npts, "obs_table_rows": w["nobs"],
                          "parallelism": f"grid-point shard x{n_gpus}" + (" + RCCL obs all-gather" if world > 1 else "")},
               "analysis_wall_s": elapsed / args.steps, "nonzero_status_points": bad,
               "jacobi_sweeps_mean": sweeps_mean, "lists": args.lists, "search_ms": search_ms,
               "search_in_step": bool(args.lists in ("search", "columns") and args.search_in_step),
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
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


def cpu_baseline(w, relax, seconds, threads=0):
    """The oracle's OpenMP restatement of the same loop body on a bounded sample of the same points (kind "port"),
    timed on this box's host cores.  A reported baseline, not the target."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    ncores = host_threads(threads)
    k, nv, npts = w["k"], w["nv"], w["npts"]
    ens = w["ensval"].cpu().numpy()
    dep = w["dep"].cpu().numpy()
    off_all = w["obs_off"].cpu().numpy()
    rng = np.random.default_rng(1)

    def sample(ns):
        pts = np.sort(rng.choice(npts, size=ns, replace=False))
        cnt = off_all[pts + 1] - off_all[pts]
        off = np.zeros(ns + 1, dtype=np.int64)
        np.cumsum(cnt, out=off[1:])
        sel = np.concatenate([np.arange(off_all[p], off_all[p + 1]) for p in pts]) if ns else np.zeros(0, np.int64)
        st = torch_index(w, sel)
        gv = w["gues"].view(nv, w["nens"], npts)[:, :, pts].contiguous().cpu().numpy().reshape(-1)
        return off, st, gv, ns

    def torch_index(w, sel):
        import torch
        s = torch.from_numpy(sel).to(w["obs_idx"].device)
        return (w["obs_idx"][s].cpu().numpy(), w["rdiag"][s].cpu().numpy(), w["rloc"][s].cpu().numpy())

    def run(ns):
        off, (idx, rd, rl), gv, ns = sample(ns)
        prm = _oracle.DasParams(k=k, nv=nv, det_run=0, infl_adaptive=0, relax_to_inflated_prior=0,
                                relax_alpha=relax.get("relax_alpha", 0.0),
                                relax_alpha_spread=relax.get("relax_alpha_spread", 0.0), q_update_top=0.0,
                                q_sprd_max=0.0, iv_p=4, iv_q_first=5, iv_q_last=10, nthreads=ncores)
        t0 = time.perf_counter()
        r = _oracle.das_points(prm, off, idx, rd, rl, ens, dep, None, np.ones(ns * nv), gv, 1, ns, ns * w["nens"])
        dt = time.perf_counter() - t0
        assert r["rc"] == 0
        return ns / dt

    rate0 = run(min(npts, 64 * ncores))
    ns = int(min(npts, max(64 * ncores, rate0 * seconds)))
    rate = run(ns)
    return {"value": rate, "unit": "solves/s", "cores": ncores, "kind": "port",
            "sample": f"{ns} randomly chosen grid points of the same workload (all {nv} variables, same relaxation), "
                      f"oracle/letkf_oracle.c orc_das_letkf_points, OpenMP dynamic over points"}


if __name__ == "__main__":
    main()
