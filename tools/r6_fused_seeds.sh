cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_fused
mkdir -p $O
export Q1_TUNABLEOP=0
run() { # mode seed
  timeout 400 python tools/train_ppo.py --iters 1300 --envs 16384 --horizon 128 --lr 3e-5 --epochs 8 --minibatch 32768 --entropy 0.01 --kl-target 0.0036 --zero-start-prob 0.1 \
      --fused-policy --resident --fused-loss --native --log-every 100 --seed $2 --out-stride 10 --step-mode $1 --out $O/r6_train_ppo_largebatch_$1_seed$2.json > $O/largebatch_$1_seed$2.log 2>&1
  echo "large-minibatch $1 seed $2: $(tail -1 $O/largebatch_$1_seed$2.log | cut -c1-200)" | tee -a $O/largebatch_modes.txt
}
for job in $JOBS; do run ${job%%:*} ${job##*:}; done
