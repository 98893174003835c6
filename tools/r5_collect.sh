#!/bin/bash
# copy what tools/r5_evidence.sh left under gpurun_out/ into profiles/ (the tracked, judged copies)
set -u
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r5; F=gpurun_out/r5_final
cp $P/summary.txt profiles/r5_summary.txt
cp $P/pmc.json profiles/pmc.json
for f in $P/*_kernel_stats.csv; do cp $f profiles/r5_$(basename $f); done
for n in 32768 262144; do cp $P/sampler_$n.txt profiles/r5_sampler_$n.txt; done
cp $F/r5_bench_driver_steps20.json $F/r5_bench_driver_steps20_extra.json $F/r5_bench_2rank_1gpu.json $F/r5_bench_8rank_1gpu.json profiles/
cp $F/r5_train_ppo_refcfg_persistent_final2.json profiles/r5_train_ppo_refcfg_persistent_v3_seed0.json
for s in 1 2; do [ -f $F/r5_train_ppo_refcfg_persistent_seed$s.json ] && cp $F/r5_train_ppo_refcfg_persistent_seed$s.json profiles/r5_train_ppo_refcfg_persistent_v3_seed$s.json; done
{ echo "# tools/r5_evidence.sh, $(cat $F/build_id.txt)"; echo "# pytest -m gpu:"; cat $F/pytest_gpu.txt; cat $F/smoke.txt; for f in time_learner_persistent time_learner_persistent_prof time_learner_persistent_agent_scope; do echo "# $f:"; cat $F/$f.json; done; } > profiles/r5_final_run.txt
ls -la profiles | wc -l
