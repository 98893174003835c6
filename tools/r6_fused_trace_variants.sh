cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6_fused
for v in "" 256 273; do
  export Q1ENV_LIB_PATH=$GRAFT_REPO_ROOT/q1physrl_amd/libq1env_fzstamps$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_fused/trv_$v -o t -- python tools/time_learner.py --phase step --steps 100 --step-mode fused_dw1 > gpurun_out/r6_fused/trv_$v.json 2> gpurun_out/r6_fused/trv_$v.err
  echo "== variant $v"; cat gpurun_out/r6_fused/trv_$v.json
  python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/r6_fused/trv_$v/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f}")
PY
done
find gpurun_out/r6_fused -name '*.csv' -size +1M -delete
