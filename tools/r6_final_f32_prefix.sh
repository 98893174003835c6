#!/bin/bash
# Round 6: the float32 learner's reference-configuration runs of profiles/r6_seed_table.md are from the round's earlier build (4f14e911c949cf65); one barrier
# of that kernel was moved afterwards.  This re-runs the first IT (default 600) iterations of seeds 0 and 1 on the FINAL build; the logged rows must equal the
# earlier build's (tools/r6_final_f32_compare.py).  One gpurun call, the runs concurrent.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_final_f32
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16(), 'lib sha16', L.lib_sha16())" > $O/build_id.txt 2>&1
for s in 0 1; do
  timeout 600 python tools/train_ppo.py --refcfg --native --fused-policy --iters ${IT:-600} --log-every 100 --eval-every 100 --out-stride 10 --seed $s \
      --learner-fp32 --out $O/r6_train_ppo_refcfg_f32_finalbuild_prefix_seed$s.json > $O/f32_seed$s.log 2>&1 &
done
wait
cat $O/build_id.txt
for s in 0 1; do echo "f32 seed $s: $(tail -1 $O/f32_seed$s.log | cut -c1-300)"; done
