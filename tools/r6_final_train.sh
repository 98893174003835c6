#!/bin/bash
# Round 6, end-to-end runs on the round's FINAL build (one gpurun call, the runs concurrent): the reference configuration with the float16 persistent
# learner, seed 0 (must end on round 5's / this round's earlier-build numbers to the last digit: 5 666 deterministic / 5 606 stochastic) and seed 8 (a
# ninth float16 seed next to the nine float32 ones), and the large-minibatch configuration in its default (automatic = fused + products) mode, seed 0.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_final_train
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16(), 'lib sha16', L.lib_sha16())" > $O/build_id.txt 2>&1
for s in 0 8; do
  timeout 1200 python tools/train_ppo.py --refcfg --native --fused-policy --iters 2989 --log-every 100 --eval-every 100 --out-stride 10 --seed $s \
      --out $O/r6_train_ppo_refcfg_f16_finalbuild_seed$s.json > $O/f16_seed$s.log 2>&1 &
done
timeout 400 python tools/train_ppo.py --iters 1300 --envs 16384 --horizon 128 --lr 3e-5 --epochs 8 --minibatch 32768 --entropy 0.01 --kl-target 0.0036 --zero-start-prob 0.1 \
    --fused-policy --resident --fused-loss --native --log-every 100 --seed 0 --out-stride 10 --out $O/r6_train_ppo_largebatch_auto_finalbuild_seed0.json > $O/largebatch_seed0.log 2>&1 &
wait
cat $O/build_id.txt
for s in 0 8; do echo "refcfg f16 seed $s: $(tail -1 $O/f16_seed$s.log | cut -c1-400)"; done
echo "large-minibatch auto seed 0: $(tail -1 $O/largebatch_seed0.log | cut -c1-300)"
