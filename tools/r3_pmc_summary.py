#!/usr/bin/env python3
"""Per-kernel summary of a tools/r3_pmc.sh directory: average launch duration (kernel stats), FETCH_SIZE / WRITE_SIZE per
launch (KB -> bytes; FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950: 128-B requests are tallied at 64 B) and
the SQ ratios.  Writes <dir>/summary.json and prints it."""
import csv, json, os, sys
from collections import defaultdict
d = sys.argv[1]
short = lambda n: n.replace("void ", "").replace("letkf::", "").split("(")[0]
out = {}
bj = os.path.join(d, "bench.json")
if os.path.exists(bj):
    try:
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        out["bench"] = {k: b.get(k) for k in ("value", "ms_per_step", "eigenfree_iterations_mean", "nonzero_status_points", "parity_sample_max_rel")}
        out["bench"]["workload"] = b["config"]["workload"]
        out["bench"]["roofline"] = {k: b["roofline"].get(k) for k in ("bound", "achieved", "peak", "frac", "kernel_ms", "launches", "alg_bytes_per_solve", "alg_flops_per_solve")}
        out["points"] = b["config"]["points_per_gpu"]
    except Exception as e:
        out["bench_error"] = str(e)
ks = {}
p = os.path.join(d, "kernel_stats.csv")
if os.path.exists(p):
    for r in csv.DictReader(open(p)):
        if "letkf::" in r["Name"]:
            ks[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
def counters(name):
    p = os.path.join(d, f"pmc_{name}.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for kn, cs in counters(cname).items():
        v = cs.get(cname)
        if v:
            ks.setdefault(kn, {})[cname + "_KB_per_launch"] = sum(v) / len(v)
for kn, cs in counters("sq").items():
    g = {c: sum(v) / len(v) for c, v in cs.items()}
    e = ks.setdefault(kn, {})
    if g.get("SQ_WAVE_CYCLES"):
        e["wait_inst_any_frac_of_wave_cycles"] = g.get("SQ_WAIT_INST_ANY", 0) / g["SQ_WAVE_CYCLES"]
        e["wait_lds_frac_of_wave_cycles"] = g.get("SQ_WAIT_INST_LDS", 0) / g["SQ_WAVE_CYCLES"]
    if g.get("GRBM_GUI_ACTIVE"):
        e["valu_busy_frac"] = g.get("SQ_ACTIVE_INST_VALU", 0) / (g["GRBM_GUI_ACTIVE"] * 32.0)   # (same method as r01 / r02: quad-cycles of 8 SEs x 4)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS"):
        if c in g:
            e[c + "_per_launch"] = g[c]
for kn, e in ks.items():
    f, w = e.get("FETCH_SIZE_KB_per_launch"), e.get("WRITE_SIZE_KB_per_launch")
    if f is not None and w is not None:
        e["traffic_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        e["traffic_bytes_per_launch_raw"] = (f + w) * 1024.0
        if e.get("avg_ms"):
            e["traffic_TB_per_s"] = e["traffic_bytes_per_launch"] / (e["avg_ms"] * 1e-3) / 1e12
out["kernels"] = ks
out["note"] = ("traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md: 128-B read requests are tallied at 64 B); "
               "these are L2 <-> fabric bytes -- Infinity-Cache hits are counted --, per launch = one batch of the staged path")
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
