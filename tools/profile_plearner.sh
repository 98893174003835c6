#!/bin/bash
# On the GPU box: rocprofv3 evidence for the persistent learner (tools/time_learner_persistent.py: 3 dispatches of 11 730 steps):
# kernel-trace statistics, then separate PMC passes (instruction mix; HBM bytes).  Output: gpurun_out/prof_plearner/.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# KERNEL=f32: the float32 kernel (tools/time_learner_f32.py with EPOCHS=30: dispatches of 11 730 steps; its float16 dispatches are filtered out by name)
if [ "${KERNEL:-f16}" = "f32" ]; then O=gpurun_out/prof_plearner_f32; CMD="env EPOCHS=30 python tools/time_learner_f32.py"; KPAT=persistent_learner_f32_kernel
else O=gpurun_out/prof_plearner; CMD="python tools/time_learner_persistent.py"; KPAT=persistent_learner_kernel; fi
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD > $O/trace.out 2> $O/trace.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/sq -o t -- $CMD > /dev/null 2> $O/sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o t -- $CMD > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o t -- $CMD > /dev/null 2> $O/write.err
python - <<PY
import csv, glob, collections
O = "$O"
KPAT = "$KPAT"
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True) + glob.glob(O + "/trace/*kernel_stats.csv"):
    print("== kernel-trace statistics (" + f + ")")
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-70s calls=%s avg_us=%.1f pct=%s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("sq", "fetch", "write"):
    for f in glob.glob(O + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KPAT in r["Kernel_Name"]:
                agg[r["Counter_Name"]][r.get("Dispatch_Id", "0")].append(float(r["Counter_Value"]))
steps = 11730
print("== " + KPAT + ", per dispatch of %d steps (counter summed over the dispatch's rows, mean over dispatches)" % steps)
for c, d in sorted(agg.items()):
    vals = [sum(v) for v in d.values()]
    m = sum(vals) / len(vals)
    extra = ""
    if c.startswith("SQ_INSTS") :
        extra = "  -> %.0f per step per wave (64 active waves: 2 groups x 8 workgroups x 4; the 48 other workgroups exit at once)" % (m / steps / 64)
    if c in ("FETCH_SIZE", "WRITE_SIZE"):
        extra = "  -> %.1f KB per step (KiB units of 32 B / 64 B requests: see MI355X_MICROARCH.md for the corrections)" % (m / steps)
    print("  %-22s %.4g%s" % (c, m, extra))
PY
