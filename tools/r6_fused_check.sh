#!/bin/bash
# On the GPU box (round 6): the fused forward + backward kernel of the large-minibatch SGD step (csrc/q1learner_fused.hpp) - its parity tests, then the
# step's time in the three kernel sequences at the large-minibatch configuration's 32 768 samples and at smaller sizes.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6_fused
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_learner.py -x -q -k "fused_forward_backward or sgd_step_equals" > $OUT/tests.log 2>&1
echo "tests rc $?" >> $OUT/tests.log
tail -5 $OUT/tests.log
for mb in 32768 8192 2048 512 128; do
  for mode in four_launch fused fused_dw1; do
    timeout 300 python tools/time_learner.py --phase step --steps 200 --mb $mb --step-mode $mode 2>> $OUT/time.err
  done
done | tee $OUT/times.jsonl
