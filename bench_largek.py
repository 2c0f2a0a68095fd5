#!/usr/bin/env python3
"""Throughput of the large-k paths (BASELINE configs C3 k=320, C5 k=1000; plus k=100/128) through
letkf_core_batch_dev on a sampled batch of problems with n=200 local obs.  Not the contract bench."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    pkg.build()
    ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
    out = {}
    for k, nb in [(64, 4096), (100, 4096), (128, 4096), (160, 4096), (320, 4096), (1000, 1024)]:
        n = 200
        g = torch.Generator(device="cuda").manual_seed(k)
        H = torch.randn(nb, k, n, generator=g, dtype=torch.float64, device="cuda")
        H -= H.mean(dim=1, keepdim=True)
        rloc = torch.rand(nb, n, generator=g, dtype=torch.float64, device="cuda") * 0.9 + 0.05
        rdiag = 9.0 / rloc
        dep = torch.randn(nb, n, generator=g, dtype=torch.float64, device="cuda")
        infl = torch.ones(nb, dtype=torch.float64, device="cuda")
        nobsl = torch.full((nb,), n, dtype=torch.int32, device="cuda")
        trans = torch.empty(nb, k * k, dtype=torch.float64, device="cuda")
        transm = torch.empty(nb, k, dtype=torch.float64, device="cuda")
        status = torch.zeros(nb, dtype=torch.int32, device="cuda")
        nsw = torch.zeros(nb, dtype=torch.int32, device="cuda")
        run = lambda: ctx.core_batch(k, n, nobsl, H, rdiag, rloc, dep, infl, trans, transm=transm, rdiag_wloc=True,
                                     status=status, nsweep=nsw)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"k{k}"] = dict(batch=nb, seconds=dt, solves_per_s=nb / dt, sweeps=float(nsw.double().mean()),
                            bad=int((status != 0).sum()))
        print(k, out[f"k{k}"], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
