#!/bin/bash
# Round 6, the evidence set of the tree as it is (one gpurun call): GPU tests + smoke, the driver's bench line (+ the 2- and 8-rank forms oversubscribed
# on the one GPU), the persistent learners' step times (float16: three exchange modes + the phase clock; float32), the policy-forward experiment.
# Everything lands under gpurun_out/r6_final/ (tools/r6_collect.sh copies what is to be judged into profiles/).  The profile round (rocprofv3 kernel-trace
# statistics + PMC passes -> profiles/pmc.json for THIS build) is its own call: tools/profile_round.sh r6 all.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_final
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16(), 'lib sha16', L.lib_sha16())" > $O/build_id.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.txt
Q1_BENCH_EXTRA=$O/r6_bench_driver_steps20_extra.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_bench_driver_steps20.json 2> $O/bench_driver.err
for n in 2 8; do
    Q1_BENCH_OVERSUBSCRIBE=1 Q1_BENCH_EXTRA=$O/extra_$n.json timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) \
        bench.py --gpus $n --steps 20 --warmup 5 > $O/r6_bench_${n}rank_1gpu.json 2> $O/bench_$n.err
done
for m in auto agent census_fail; do MODE=$m timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > $O/time_learner_persistent_$m.json; done
PROF=0 timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > $O/time_learner_persistent_prof0.json
for m in auto agent; do MODE=$m timeout 300 python tools/time_learner_f32.py 2>&1 | tail -1 > $O/time_learner_f32_$m.json; done
PROF=0 EPOCHS=4 timeout 300 python tools/time_learner_f32.py 2>&1 | tail -1 > $O/time_learner_f32_prof0.json
timeout 300 python tools/r6_f32_err.py 2>&1 | tail -14 > $O/f32_grad_error.txt
bash tools/r6_policy_tanh.sh > $O/policy_tanh.txt 2>&1
cat $O/build_id.txt $O/pytest_gpu.txt $O/smoke.txt; cut -c1-600 $O/r6_bench_driver_steps20.json; for f in $O/time_learner_*.json; do echo $f; cut -c1-260 $f; done; cat $O/policy_tanh.txt
