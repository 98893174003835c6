"""Print a rocprofv3 kernel_stats.csv as a short table (kernel, calls, avg us, share)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r["Name"].replace("void ", "").split("(")[0]
    print(f"{name[:90]:90s} calls={int(r['Calls']):6d} avg_us={float(r['AverageNs']) / 1e3:9.2f} pct={float(r['Percentage']):6.2f}")
