#!/bin/bash
# Round 6: more seeds of the float16 persistent learner under the reference configuration (the round-5 runs had seeds 0 - 4), same command as
# tools/r6_train_f32.sh without --learner-fp32.   $1.. seeds (concurrent)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_train
mkdir -p $O
export Q1_TUNABLEOP=0
for s in "$@"; do
  timeout ${TMO:-1500} python tools/train_ppo.py --refcfg --native --fused-policy --iters ${IT:-2989} --log-every 100 --eval-every 100 --out-stride 10 --seed $s \
      --out $O/r6_train_ppo_refcfg_f16_seed$s.json > $O/f16_seed$s.log 2>&1 &
done
wait
for s in "$@"; do echo "f16 seed $s: $(tail -1 $O/f16_seed$s.log | cut -c1-300)"; done
