#!/bin/bash
# Round 6, VERDICT r5 item 3: the fp32 CONTROL of the reference-configuration training run - the same run (data/params.yml + RLlib 0.8.4 defaults,
# 2 989 iterations = 149.45 M env-steps, same seeds, same permutations) with the learner in float32 arithmetic (q1env_learner_sgd_epochs_f32).
#   $1.. seeds (run concurrently on the one GPU: a learner occupies 16 of the 256 CUs; three at a time leave CUs free on every XCD for the samplers)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_train
mkdir -p $O
export Q1_TUNABLEOP=0
for s in "$@"; do
  timeout ${TMO:-3300} python tools/train_ppo.py --refcfg --native --fused-policy --iters ${IT:-2989} --log-every 100 --eval-every 100 --out-stride 10 --seed $s \
      --learner-fp32 ${EXTRA:-} --out $O/r6_train_ppo_refcfg_f32_seed$s.json > $O/f32_seed$s.log 2>&1 &
done
wait
for s in "$@"; do echo "seed $s: $(tail -1 $O/f32_seed$s.log | cut -c1-300)"; done
