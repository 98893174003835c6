#!/bin/bash
# copy what the round-6 GPU calls left under gpurun_out/ into profiles/ (the tracked, judged copies)
set -u
cd "$(dirname "$0")/.."
T=gpurun_out/r6_train; F=gpurun_out/r6_final; P=gpurun_out/prof_r6
for f in $T/r6_train_ppo_refcfg_f32_seed*.json $T/r6_train_ppo_refcfg_f16_seed*.json; do [ -f $f ] && cp $f profiles/; done
if [ -d $F ]; then
  for f in r6_bench_driver_steps20.json r6_bench_driver_steps20_extra.json r6_bench_2rank_1gpu.json r6_bench_8rank_1gpu.json; do [ -f $F/$f ] && cp $F/$f profiles/; done
  { echo "# tools/r6_evidence.sh, $(cat $F/build_id.txt)"; echo "# pytest -m gpu:"; cat $F/pytest_gpu.txt; cat $F/smoke.txt
    for f in $F/time_learner_*.json; do echo "# $(basename $f .json):"; cat $f; done
    echo "# gradient error against float64 torch autograd (tools/r6_f32_err.py):"; cat $F/f32_grad_error.txt; } > profiles/r6_final_run.txt
  [ -f $F/policy_tanh.txt ] && cp $F/policy_tanh.txt profiles/r6_policy_tanh_gpu.txt
fi
if [ -d $P ]; then
  cp $P/summary.txt profiles/r6_summary.txt
  cp $P/pmc.json profiles/pmc.json
  for f in $P/*_kernel_stats.csv; do cp $f profiles/r6_$(basename $f); done
  for n in 32768 262144; do [ -f $P/sampler_$n.txt ] && cp $P/sampler_$n.txt profiles/r6_sampler_$n.txt; done
fi
python tools/r6_seed_table.py > profiles/r6_seed_table.md
cat profiles/r6_seed_table.md
