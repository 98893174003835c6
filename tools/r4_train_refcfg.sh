#!/bin/bash
# round 4: PPO with the REFERENCE's training configuration, native learner for all 2 989 iterations of the
# reference run, then the float32 torch learner (hipGraph-captured, same configuration) for the first iterations as the A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_train
mkdir -p $O
timeout 3100 python tools/train_ppo.py --refcfg --native --fused-policy --iters ${1:-2989} --log-every 50 --eval-every 100 --seed 0 \
    --out $O/r4_train_ppo_refcfg_native.json --save $O/r4_policy_refcfg_native.npz > $O/native.log 2>&1
echo "native rc=$?" >> $O/native.log
timeout 700 python tools/train_ppo.py --refcfg --iters ${2:-40} --log-every 5 --eval-every 20 --seed 0 \
    --out $O/r4_train_ppo_refcfg_torch_fp32.json > $O/torch_fp32.log 2>&1
echo "torch rc=$?" >> $O/torch_fp32.log
tail -4 $O/native.log | cut -c1-400; tail -3 $O/torch_fp32.log | cut -c1-400
