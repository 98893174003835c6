#!/bin/bash
# round 4, probe 4: where the native learner's SGD step spends its time, at the training minibatch (32 768) and at the reference's (128)
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_probe4
mkdir -p $O
for cfg in "32768 32 50" "128 4 300" "128 32 300"; do
  set -- $cfg
  python tools/time_learner.py --mb $1 --splits $2 --steps $3 > $O/time_mb$1_s$2.json 2> $O/time_mb$1_s$2.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_mb$1_s$2 -o t -- python tools/time_learner.py --mb $1 --splits $2 --steps $3 --phase step > /dev/null 2> $O/trace_mb$1_s$2.err
  for f in $O/trace_mb$1_s$2/*/*kernel_stats.csv $O/trace_mb$1_s$2/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" $O/kernel_stats_mb$1_s$2.csv; done
  echo "== mb $1 splits $2"; cat $O/time_mb$1_s$2.json; python tools/kernel_stats_table.py $O/kernel_stats_mb$1_s$2.csv 2>/dev/null | head -14 || head -12 $O/kernel_stats_mb$1_s$2.csv
done
timeout 300 python tools/train_ppo.py --refcfg --native --fused-policy --iters 6 --log-every 1 2>&1 | grep iter_s | tail -3 | cut -c1-300
