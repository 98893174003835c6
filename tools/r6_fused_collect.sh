#!/bin/bash
# copy what tools/r6_fused_evidence.sh / r6_fused_seeds.sh left under gpurun_out/r6_fused/ into profiles/ (the tracked, judged copies)
set -u
cd "$(dirname "$0")/.."
F=gpurun_out/r6_fused
OUT=profiles/r6_learner_fused.txt
{
cat <<'TXT'
# Round 6: the 32 768-sample SGD step of the large-minibatch configuration ("<= 85 us / <= 350 MB"), one MI355X.
# q1env_learner_sgd_step in its kernel sequences (include/q1env.h q1env_learner_set_step_mode; csrc/q1learner_fused.hpp):
#   four_launch  round 4's forward | backward (+ in-kernel PPO loss) | split-K weight gradients | reduction + Adam + images
#   fused        forward + loss gradient + data gradients as ONE kernel, dZ1 / tanh(H2) stored as before: bit-identical to four_launch
#   fused_dw1    ... with dW1 / db1 / dW3 taken per 32-sample tile inside that kernel (no dZ1, no tanh(H2) arrays) and the shared-operand
#                weight-gradient kernel: the automatic choice from 2 048 samples on
#   fused_dw1_r4wgrad / fused_dw1_q   measurement only: round 4's weight-gradient kernel / its column-quarter form on the product arrays
# History of the round (all on fresh leases, 200 eager steps, us per step): start 98.8 | fused kernel, first version 94.7 (kernel 61.5 against forward 23.5 +
# backward 40.4) | dW1 products 95.1 | dW3 products 95.5 (442 -> 322 MB, no time: the kernel was not bound by its bytes) | saturation report per workgroup
# instead of per wave 83.2 (kernel 49.9; four_launch 93.4) | shared-operand weight-gradient kernel 81.6 | only the mapped slots of the small products written 81.1.
# What the diagnostic build's switches measured (tools/exp_fused_stamps.py -DQ1_FZ_EXP=..., fused_dw1, kernel time under rocprofv3 / latest wave end):
#   as shipped before the atomics fix 63.8 us / 57.0    no saturation report 49.95 / 45.6    no stores at all + no report 41.4 / 38.8
#   no stores at all WITH the report 59.7 / 38.8 (gap to the next dispatch 20.4 us: 2 x 2 048 atomics on two words)
#   backward stores non-temporal 68.0 (wave end)    tanh(H1) stores sc1 / nt / sc0 sc1: latest wave end 41.8 / 44.6 / 41.5, kernel unchanged
#   neighbouring XCDs swap their tiles: the same physical XCDs stay the slow ones (two odd-numbered XCDs per box: 1 + 7, 3 + 5, ...)
#   hardware exp / log in the loss: loss phase 4.4 -> 2.9 us per wave, kernel unchanged at the time (hidden behind the report's atomics); not shipped:
#   the loss stays the four-launch step's, bit for bit
TXT
echo "# ---- $(cat $F/build_id.txt 2>/dev/null)"
echo "# ---- step time per kernel sequence (tools/time_learner.py --phase step --steps 200 --step-mode M)"
cat $F/times.jsonl 2>/dev/null
for m in fused_dw1 four_launch; do echo "# ---- HBM bytes per kernel, $m (tools/profile_learner_bytes.sh r6 $m: rocprofv3 kernel-trace + FETCH_SIZE x 2 + WRITE_SIZE, separate passes)"; cat $F/bytes_$m.txt 2>/dev/null; done
echo "# ---- phase stamps of the fused kernel (tools/exp_fused_stamps.py --dw1: the diagnostic build, every wave of both networks)"
grep -v "amdgpu.ids" $F/stamps_dw1.txt 2>/dev/null
echo "# ---- large-minibatch configuration, whole training runs (665 600 SGD steps each), 16 seeds x {four_launch, fused_dw1} + fused seeds 0, 1: profiles/r6_largebatch_seed_table.md"
echo "# ---- AddressSanitizer job of the host side (tools/asan_check.sh; learner sections first, tick-server sections opt-in):"
tail -9 gpurun_out/r6_asan.log 2>/dev/null
} > $OUT
wc -l $OUT
