#!/bin/bash
# On the GPU box (round 6): per-kernel times of the 32 768-sample SGD step in the three kernel sequences (rocprofv3 --kernel-trace --stats).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6_fused
mkdir -p $OUT
for mode in ${MODES:-four_launch fused fused_dw1}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -o t -- python tools/time_learner.py --phase step --steps 100 --step-mode $mode > $OUT/trace_$mode.json 2> $OUT/trace_$mode.err
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace_$mode/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("== $mode")
    for r in rows[:8]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
done | tee $OUT/trace_summary.txt
find $OUT -name '*.csv' -size +1M -delete
