#!/bin/bash
# Round 6: rocprofv3 --kernel-trace --stats of the DRIVER'S OWN bench command on the final build (the profile round, tools/profile_round.sh, traces the
# 720-tick steady-state launches; this is the 20-tick launch the contract line's roofline is quoted on).  One gpurun call.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_driver_trace
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16(), 'lib sha16', L.lib_sha16())" > $O/build_id.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_under_profiler.json 2> $O/trace.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_unprofiled.json 2> /dev/null
python - "$O" <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
tr = sorted(glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True))
st = sorted(glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True))
out = open(O + "/summary.txt", "w")
def p(*a):
    print(*a); print(*a, file=out)
p("#", open(O + "/build_id.txt").read().strip())
p("# rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5")
for name in ("bench_line_under_profiler.json", "bench_line_unprofiled.json"):
    try:
        d = json.loads(open(O + "/" + name).read().strip().splitlines()[-1])
        p("# %s: value %.4g %s, ms_per_step %.6g, roofline %s, timed_region_us %s" % (name, d["value"], d["unit"], d["ms_per_step"],
          json.dumps({k: d["roofline"].get(k) for k in ("kernel", "avg_launch_us", "ticks_per_launch", "achieved", "frac", "traffic", "pmc_stale")}),
          json.dumps(d.get("timed_region_us"))))
    except Exception as ex:      # noqa: BLE001
        p("# %s: unreadable (%r)" % (name, ex))
if st:
    p("# kernel statistics (rocprofv3 --stats), by total time:")
    rows = list(csv.DictReader(open(st[0])))
    for r in rows[:12]:
        p("   %-72s calls %6s  avg %10.1f ns  min %8s  max %8s  %5s %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
if tr:
    rows = list(csv.DictReader(open(tr[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    roll = [r for r in rows if "rollout_kernel" in r["Kernel_Name"]]
    t0 = int(rows[0]["Start_Timestamp"])
    p("# every rollout_kernel dispatch of the run, chronological (start relative to the first kernel of the process; duration = End - Start):")
    for r in roll:
        p("   %12.1f us  %-64s %8d ns   grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:64], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", "")))
PY
