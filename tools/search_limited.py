#!/usr/bin/env python3
"""Timing probe: obs_local with MAX_NOBS_PER_GRID on the C2 grid -- two radar ctypes (REF, Vr) on the C2 lattice, each
limited to 100 observations per point (SURVEY.md section 8(d) variant: n = 200 exactly)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package
import bench_workload as bw

pkg = load_package(); pkg.build()
dev = torch.device("cuda:0")
ctx = pkg.Context(0, torch.cuda.current_stream().cuda_stream)
w = bw.build("C2", dev)
t, keep, order, pts = bw.search_tables(w, pkg, dev)
i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
d64 = lambda v: torch.tensor(v, dtype=torch.float64, device=dev)
nrow = keep["ob_ri"].numel()
# second ctype = same lattice again (rows appended, its own mesh prefix sums shifted by nrow)
ac2 = keep["ac_ext"] + nrow
k2 = dict(group_start=i32([0, 1, 2]), group_member=i32([0, 1]), vmode=i32([1, 1]),
          hori_loc=keep["hori_loc"].repeat(2), vert_loc=keep["vert_loc"].repeat(2), varloc=d64([1.0, 1.0]),
          max_nobs=i32([int(sys.argv[1]) if len(sys.argv) > 1 else 100] * 2), ngrd_i=keep["ngrd_i"].repeat(2),
          ngrd_j=keep["ngrd_j"].repeat(2), ngrdsch_i=keep["ngrdsch_i"].repeat(2), ngrdsch_j=keep["ngrdsch_j"].repeat(2),
          ngrdext_i=keep["ngrdext_i"].repeat(2), ngrdext_j=keep["ngrdext_j"].repeat(2),
          ac_off=torch.tensor([0, keep["ac_ext"].numel()], dtype=torch.int64, device=dev),
          ac_ext=torch.cat([keep["ac_ext"], ac2]).contiguous(),
          ob_ri=keep["ob_ri"].repeat(2), ob_rj=keep["ob_rj"].repeat(2), ob_lev=keep["ob_lev"].repeat(2),
          ob_dat=keep["ob_dat"].repeat(2), ob_err=keep["ob_err"].repeat(2))
t.nctype, t.ngroup = 2, 2
for k_, v in k2.items():
    setattr(t, k_, v.data_ptr())
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    off, idx, rd, rl = ctx.obs_search(t, *pts)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
cnt = (off[1:] - off[:-1])
print(f"limited search (2 ctypes x max {int(k2['max_nobs'][0])}): {ms:.1f} ms, lists {int(off[-1])} entries, per point min/mean/max {int(cnt.min())}/{float(cnt.double().mean()):.1f}/{int(cnt.max())}")
