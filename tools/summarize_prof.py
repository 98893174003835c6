#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of tools/profile_r1.sh: per-kernel stats and PMC byte counters per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def first(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


st = first("trace/**/*kernel_stats.csv")
if st:
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", st)
    for row in csv.DictReader(open(st)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
kt = first("trace/**/*kernel_trace.csv")
if kt:
    rows = list(csv.DictReader(open(kt)))
    by = defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    print("== kernel trace: per-kernel avg duration and avg gap to the previous kernel end (same queue order)")
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gaps = defaultdict(list)
    for a, b in zip(rows[:-1], rows[1:]):
        gaps[b["Kernel_Name"]].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    for k, v in by.items():
        d = [e - s for s, e in v]
        g = gaps.get(k, [0])
        g2 = sorted(g)
        print(f"{k[:70]:70s} calls={len(v):6d} avg={sum(d)/len(d):9.1f} ns  min={min(d)} max={max(d)}  median_gap={g2[len(g2)//2]} ns")
    r0 = rows[0]
    print("VGPR/SGPR/LDS of", r0["Kernel_Name"][:40], {k: r0[k] for k in r0 if "GPR" in k or "LDS" in k or "Workgroup" in k or "Grid" in k})
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = first(f"{tag}/**/*counter_collection.csv")
    if not cc:
        print("no counter csv for", tag)
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") == ctr:
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    print(f"== {ctr} (raw counter units, per launch)")
    for k, (s, c) in acc.items():
        print(f"{k[:70]:70s} launches={c:6d} avg_raw={s/c:14.2f}  -> x1024 B = {s/c*1024/1e6:10.3f} MB/launch")
