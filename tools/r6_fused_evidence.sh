#!/bin/bash
# Round 6, evidence for the fused large-minibatch SGD step (csrc/q1learner_fused.hpp): the five-seed regression of the large-minibatch configuration (now on
# q1env_learner_sgd_step's automatic mode = the fused kernel with per-tile dW1 / dW3 products), the step's time in every kernel sequence, per-kernel times and
# HBM bytes (rocprofv3), the phase stamps of the fused kernel.  Everything under gpurun_out/r6_fused/ (copied to profiles/ by hand: profiles/r6_learner_fused.txt).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_fused
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16(), 'lib sha16', L.lib_sha16())" > $O/build_id.txt 2>&1
: > $O/largebatch.txt
for s in ${SEEDS:-0 1 2 3 4}; do
  timeout 400 python tools/train_ppo.py --iters 1300 --envs 16384 --horizon 128 --lr 3e-5 --epochs 8 --minibatch 32768 --entropy 0.01 --kl-target 0.0036 --zero-start-prob 0.1 \
      --fused-policy --resident --fused-loss --native --log-every 100 --seed $s --out-stride 10 --out $O/r6_train_ppo_largebatch_fused_seed$s.json > $O/largebatch_seed$s.log 2>&1
  echo "large-minibatch seed $s: $(tail -1 $O/largebatch_seed$s.log | cut -c1-220)" >> $O/largebatch.txt
done
cat $O/build_id.txt $O/largebatch.txt
if [ "${FULL:-1}" = "1" ]; then
  for m in four_launch fused fused_dw1 fused_dw1_r4wgrad fused_dw1_q auto; do timeout 300 python tools/time_learner.py --phase step --steps 200 --step-mode $m; done > $O/times.jsonl 2>> $O/time.err
  cat $O/times.jsonl
  for m in fused_dw1 four_launch; do bash tools/profile_learner_bytes.sh r6 $m > $O/bytes_$m.txt 2>&1; cat $O/bytes_$m.txt; done
  timeout 300 python tools/exp_fused_stamps.py --dw1 > $O/stamps_dw1.txt 2>&1; cat $O/stamps_dw1.txt
fi
