#!/usr/bin/env python3
"""Timing probe: the column search with MAX_NOBS_PER_GRID on the C2 grid (two radar ctypes limited to N each), no solve."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package
import bench_workload as bw

pkg = load_package()
dev = torch.device("cuda:0")
ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
w = bw.build("C2", dev)
t, keep, order, pts = bw.search_tables(w, pkg, dev, max_nobs=int(sys.argv[1]) if len(sys.argv) > 1 else 100)
nij = w["cfg"]["nx"] * w["cfg"]["ny"]
rig, rjg = pts[0][:nij].contiguous(), pts[1][:nij].contiguous()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    off, idx, rd, rl = ctx.obs_search_columns(t, nij, w["cfg"]["nz"], rig, rjg, pts[2], pts[3])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
print(f"limited column search: {ms:.1f} ms, {int(off[-1])} entries")
