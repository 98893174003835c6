#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of tools/profile_round.sh: per-kernel stats, launch gaps, PMC byte counters per launch
(raw and with the gfx950 FETCH_SIZE x2 correction), and the calibration of both counters on the known-bytes copy kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def first(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def short(name):
    return name.replace("void ", "").split("(")[0][:60]


st = first("trace/**/*kernel_stats.csv")
if st:
    print("== rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
    for row in csv.DictReader(open(st)):
        print(f"  {short(row['Name']):60s} calls={row['Calls']:>6s} avg_ns={float(row['AverageNs']):10.1f} min={row['MinNs']:>7s} max={row['MaxNs']:>8s} pct={row['Percentage']}")
sv = first("trace_step/**/*kernel_stats.csv")
if sv:
    print("== rocprofv3 --kernel-trace --stats, bench.py --mode step (one step_kernel launch per tick)")
    for row in csv.DictReader(open(sv)):
        if float(row["Percentage"]) > 0.05:
            print(f"  {short(row['Name']):60s} calls={row['Calls']:>6s} avg_ns={float(row['AverageNs']):12.1f} min={row['MinNs']:>9s} max={row['MaxNs']:>9s} pct={row['Percentage']}")
kt = first("trace/**/*kernel_trace.csv")
if kt:
    rows = list(csv.DictReader(open(kt)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by, grid = defaultdict(list), {}
    for r in rows:
        by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        grid[r["Kernel_Name"]] = {k: r[k] for k in ("Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size") if k in r}
    print("== kernel trace (per dispatch)")
    for k, d in by.items():
        d2 = sorted(d)
        print(f"  {short(k):60s} n={len(d):6d} avg={sum(d)/len(d):9.1f} ns median={d2[len(d2)//2]} ns  {grid[k]}")


def pmc(tag, ctr):
    cc = first(f"{tag}/**/*counter_collection.csv")
    acc = defaultdict(lambda: [0.0, 0, 0])
    if not cc:
        return acc
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") == ctr:
            a = acc[(r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


res = {}
for label, ftag, wtag in (("bench", "pmc_fetch", "pmc_write"), ("bench --mode step", "pmc_fetch_step", "pmc_write_step"),
                          ("calibration", "cal_fetch", "cal_write")):
    f, w = pmc(ftag, "FETCH_SIZE"), pmc(wtag, "WRITE_SIZE")
    print(f"== PMC per launch ({label}); FETCH_SIZE / WRITE_SIZE are in KiB; corrected fetch = raw x2 (gfx950, MI355X_MICROARCH.md HBM section)")
    for key in sorted(set(f) | set(w), key=str):
        name, gsz = key
        fr = f[key][0] / f[key][1] if key in f and f[key][1] else float("nan")
        wr = w[key][0] / w[key][1] if key in w and w[key][1] else float("nan")
        print(f"  {short(name):52s} grid={gsz:9d} launches={f[key][1] if key in f else 0:5d}  fetch_raw={fr*1024/1e6:9.3f} MB  fetch_x2={2*fr*1024/1e6:9.3f} MB  write={wr*1024/1e6:9.3f} MB"
              f"   per-thread: fetch_x2={2*fr*1024/max(gsz,1):7.2f} B write={wr*1024/max(gsz,1):7.2f} B")
        res[f"{label}:{short(name)}:{gsz}"] = {"fetch_raw_B": fr * 1024, "fetch_x2_B": 2 * fr * 1024, "write_B": wr * 1024, "threads": gsz}
json.dump(res, open(os.path.join(out, "pmc.json"), "w"), indent=1)
