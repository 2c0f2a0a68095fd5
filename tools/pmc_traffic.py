#!/usr/bin/env python3
"""profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from the two PMC passes of tools/profile.sh.
Usage: tools/pmc_traffic.py TAG "build note"   (reads gpurun_out/prof_TAG/pmc_{FETCH,WRITE}_SIZE.csv)"""
import csv, json, os, sys
tag, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kb(name):
    rows = list(csv.DictReader(open(os.path.join(root, "gpurun_out", f"prof_{tag}", f"pmc_{name}.csv"))))
    rows = [r for r in rows if "letkf_wave_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    assert rows, name
    return sum(float(r["Counter_Value"]) for r in rows) / len(rows), rows[0]["Kernel_Name"]
f, kn = kb("FETCH_SIZE")
w, _ = kb("WRITE_SIZE")
old = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
out = {"workload": "C2", "kernel": kn.replace("void letkf::", "").replace("(letkf::PointArgs)", ""), "build": note,
       "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch_raw": (f + w) * 1024.0,
       "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile.sh; the csv rows are under profiles/ "
               "beside the kernel stats and the SQ pass of the same build), one launch = 3,456,000 solves.  hbm_bytes_per_launch "
               "applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B: doubled); the "
               "gathers here are 8 B/lane in 128-B segments, a width the guide calls uncalibrated, so the raw sum is kept beside "
               "it.  These are L2<->fabric bytes: Infinity-Cache hits are counted, so this is an upper bound on HBM traffic.  "
               "Algorithmic bytes per launch: 328 GB (95.0 KB x 3.456 M).  The warm-start workspace is 2 x 88 GB of it by design "
               "(25.6 KB store + load of the previous point's eigenvectors per solve, served by the Infinity Cache at best); the "
               "rest of the excess is register spill traffic and the reference's point-fastest state layout (8-byte accesses "
               "npts*8 B apart; pmc_traffic_member.json has the member-fastest layout).",
       "history": old.get("history", []) + [{"build": old.get("build"), "FETCH_SIZE_KB": old.get("FETCH_SIZE_KB"),
                                             "WRITE_SIZE_KB": old.get("WRITE_SIZE_KB")}]}
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_per_launch")}))
