#!/bin/bash
# Round 5: the visibility tests of the completion signal, the same probe against the other store forms (controls), and the
# driver-shaped bench line with each form / prefetch depth.  Output: gpurun_out/r5_vis/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r5_vis; mkdir -p $O; rm -f $O/summary.txt
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_signal.py -x -q > $O/test_hip_signal.log 2>&1; echo "signal tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/test_hip_signal.log | tee -a $O/summary.txt
for lib in "" $(ls tools/_bin/libq1env_vis_*.so 2>/dev/null); do
  tag=${lib:-product}; tag=$(basename $tag .so)
  env ${lib:+Q1ENV_LIB_PATH=$PWD/$lib} timeout 600 python tools/visibility_probe.py --reps 400 > $O/probe_$tag.json 2> $O/probe_$tag.err; echo "probe $tag rc=$?" | tee -a $O/summary.txt
  cat $O/probe_$tag.json | tee -a $O/summary.txt
  for depth in 0 1 2; do
   for rep in 1 2; do
    t=${tag}_d${depth}_$rep
    env Q1_BENCH_EXTRA=$O/bench_extra_$t.json ${lib:+Q1ENV_LIB_PATH=$PWD/$lib} $( [ $depth != 0 ] && echo Q1ENV_ROLLOUT_DEPTH=$depth ) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$t.json 2> $O/bench_$t.err
    echo "bench $t rc=$? bytes=$(wc -c < $O/bench_$t.json)" | tee -a $O/summary.txt
    python - <<PY | tee -a $O/summary.txt
import json
d=json.load(open("$O/bench_$t.json"))
print({k:d.get(k) for k in ("value","ms_per_step_incl_runtime_sync","timed_region_us","steady_state_us_per_tick")})
PY
   done
  done
done
