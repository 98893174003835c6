#!/bin/bash
# round 5, final evidence set of the tree as it is (one gpurun call): GPU tests + smoke, the driver's bench line (+ the 2- and 8-rank forms
# oversubscribed on the one GPU), the profile round (kernel-trace statistics + PMC passes of every BASELINE config), the persistent learner's
# timing, the reference-configuration training run.  Everything lands under gpurun_out/r5_final/ (copy what is to be judged into profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_final
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16())" > $O/build_id.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.txt
Q1_BENCH_EXTRA=$O/r5_bench_driver_steps20_extra.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5_bench_driver_steps20.json 2> $O/bench_driver.err
for n in 2 8; do
    Q1_BENCH_OVERSUBSCRIBE=1 Q1_BENCH_EXTRA=$O/extra_$n.json timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) \
        bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline > $O/r5_bench_${n}rank_1gpu.json 2> $O/bench_$n.err
done
timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > $O/time_learner_persistent.json
PROF=0 timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > $O/time_learner_persistent_prof.json
MODE=agent timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > $O/time_learner_persistent_agent_scope.json
timeout 1300 python tools/train_ppo.py --refcfg --native --fused-policy --iters 2989 --log-every 50 --eval-every 100 --seed 0 --checkpoint-dir /tmp/r5_ck_final \
    --out $O/r5_train_ppo_refcfg_persistent_final2.json --save $O/r5_policy_refcfg_final2.npz > $O/train.log 2>&1
tail -2 $O/train.log | cut -c1-300 > $O/train_tail.txt
timeout 3000 bash tools/profile_round.sh r5 all > $O/profile_round.log 2>&1
for sd in ${SEEDS:-}; do
    timeout 1300 python tools/train_ppo.py --refcfg --native --fused-policy --iters 2989 --log-every 50 --eval-every 100 --seed $sd --checkpoint-dir /tmp/r5_ck_s$sd \
        --out $O/r5_train_ppo_refcfg_persistent_seed$sd.json > $O/train_s$sd.log 2>&1
    tail -1 $O/train_s$sd.log | cut -c1-200 >> $O/train_tail.txt
done
cat $O/build_id.txt $O/pytest_gpu.txt $O/smoke.txt $O/train_tail.txt; cut -c1-400 $O/r5_bench_driver_steps20.json; cut -c1-200 $O/time_learner_persistent.json
