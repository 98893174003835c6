#!/bin/bash
# round 5: PPO with the REFERENCE's training configuration (data/params.yml + RLlib 0.8.4 defaults), the persistent learner (one dispatch
# per update) for all 2 989 iterations of the reference run, with the reference's checkpoint schedule.
#   $1 iterations (2989)   $2 tag (dynscale)   $3 timeout s (2400)   $4.. extra train_ppo.py flags (e.g. --static-loss-scale, --no-persistent)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_train
mkdir -p $O
export Q1_TUNABLEOP=0
IT=${1:-2989}; TAG=${2:-dynscale}; TMO=${3:-2400}; shift 3 2>/dev/null || true
timeout $TMO python tools/train_ppo.py --refcfg --native --fused-policy --iters $IT --log-every 50 --eval-every 100 --seed 0 \
    --dynamic-loss-scale --checkpoint-dir /tmp/r5_ck_$TAG --out $O/r5_train_ppo_refcfg_$TAG.json --save $O/r5_policy_refcfg_$TAG.npz "$@" > $O/$TAG.log 2>&1
echo "$TAG rc=$?" >> $O/$TAG.log
tail -4 $O/$TAG.log | cut -c1-700
