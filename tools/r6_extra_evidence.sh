#!/bin/bash
# Round 6, second evidence call: rocprofv3 evidence for both persistent learner kernels, the five-seed large-minibatch regression (the four-launch path is
# untouched this round: must reproduce round 5's 5 652 - 5 781 in ~73 s each), the AddressSanitizer job of the host side (new entry points).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_extra
mkdir -p $O
export Q1_TUNABLEOP=0
bash tools/profile_plearner.sh > $O/plearner_rocprof_f16.txt 2>&1
KERNEL=f32 bash tools/profile_plearner.sh > $O/plearner_rocprof_f32.txt 2>&1
for s in 0 1 2 3 4; do
  timeout 400 python tools/train_ppo.py --iters 1300 --envs 16384 --horizon 128 --lr 3e-5 --epochs 8 --minibatch 32768 --entropy 0.01 --kl-target 0.0036 --zero-start-prob 0.1 \
      --fused-policy --resident --fused-loss --native --log-every 100 --seed $s --out-stride 10 --out $O/r6_train_ppo_largebatch_seed$s.json > $O/largebatch_seed$s.log 2>&1
  echo "large-minibatch seed $s: $(tail -1 $O/largebatch_seed$s.log | cut -c1-220)" >> $O/largebatch.txt
done
bash tools/asan_check.sh > $O/asan.txt 2>&1; echo "asan rc=$?" >> $O/asan.txt
tail -12 $O/plearner_rocprof_f16.txt; tail -12 $O/plearner_rocprof_f32.txt; cat $O/largebatch.txt; tail -4 $O/asan.txt
