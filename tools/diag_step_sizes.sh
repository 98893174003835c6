#!/bin/bash
# Counters behind the per-tick step kernel's large-batch behaviour: L2 hit rate, fabric reads, L1->L2 latency,
# TLB misses and wave-level stall accounting of step_kernel and of the known-bytes copy kernel at 262 144 / 1 M / 4 M envs.
# One counter group per rocprofv3 pass (only --kernel-trace next to --pmc).  Output: gpurun_out/diag_step/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/diag_step
mkdir -p $OUT
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o t -- python tools/diag_step_sizes.py > /dev/null 2> $OUT/p$i.err
done
python tools/diag_step_sizes.py --summarize $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.csv' -size +1M -delete
