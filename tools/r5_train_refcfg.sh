#!/bin/bash
# round 5: PPO with the REFERENCE's training configuration, the persistent learner (one dispatch per update) for all 2 989 iterations of
# the reference run, with the reference's checkpoint schedule; then the four-launch step for the first iterations as the A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_train
mkdir -p $O
export Q1_TUNABLEOP=0
timeout ${3:-2400} python tools/train_ppo.py --refcfg --native --fused-policy --iters ${1:-2989} --log-every 50 --eval-every 100 --seed 0 \
    --checkpoint-dir /tmp/r5_ck --out $O/r5_train_ppo_refcfg_persistent.json --save $O/r5_policy_refcfg_persistent.npz > $O/persistent.log 2>&1
echo "persistent rc=$?" >> $O/persistent.log
timeout 300 python tools/train_ppo.py --refcfg --native --fused-policy --no-persistent --iters ${2:-40} --log-every 5 --eval-every 20 --seed 0 \
    --out $O/r5_train_ppo_refcfg_fourlaunch.json > $O/fourlaunch.log 2>&1
echo "four-launch rc=$?" >> $O/fourlaunch.log
tail -4 $O/persistent.log | cut -c1-500; tail -3 $O/fourlaunch.log | cut -c1-500
