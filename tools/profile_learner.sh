#!/bin/bash
# On the GPU box: per-kernel statistics and SQ counters of the native learner's kernels (tools/time_learner.py), one counter group per pass.
set -u
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_learner_$TAG
mkdir -p $OUT
CMD="python tools/time_learner.py --steps 30"
$CMD > $OUT/unprofiled.json 2> $OUT/unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > /dev/null 2> $OUT/trace.err
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
SQ3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32"
rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/sq1 -o t -- $CMD > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/sq2 -o t -- $CMD > /dev/null 2> $OUT/sq2.err
rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $OUT/sq3 -o t -- $CMD > /dev/null 2> $OUT/sq3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o t -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o t -- $CMD > /dev/null 2> $OUT/write.err
python - <<PY
import csv, glob, collections, json
out = "$OUT"
print(open(out + "/unprofiled.json").read())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("sq1", "sq2", "sq3", "fetch", "write"):
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("q1learn::", "").replace("void ", "")
            if "learner" in k or "ppo_loss" in k or "adam" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "learn" in r["Name"] or "ppo_loss" in r["Name"]:
            print(r["Name"].split("(")[0][:60].ljust(62), "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"])
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print("--", k)
    print("   ", "  ".join(f"{n}={v:.4g}" for n, v in sorted(m.items())))
    if "SQ_WAVES" in m and m["SQ_WAVES"] > 0:
        w = m["SQ_WAVES"]
        print(f"    per wave: cycles={4 * m.get('SQ_WAVE_CYCLES', 0) / w:.0f}  VALU insts={m.get('SQ_INSTS_VALU', 0) / w:.0f}  VALU busy={4 * m.get('SQ_ACTIVE_INST_VALU', 0) / w:.0f}"
              f"  MFMA insts={m.get('SQ_INSTS_MFMA', 0) / w:.0f}  MFMA busy cycles={4 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w:.0f}  LDS insts={m.get('SQ_INSTS_LDS', 0) / w:.0f}"
              f"  wait_any={4 * m.get('SQ_WAIT_ANY', 0) / w:.0f}  wait_inst_any={4 * m.get('SQ_WAIT_INST_ANY', 0) / w:.0f}  wait_lds={4 * m.get('SQ_WAIT_INST_LDS', 0) / w:.0f}")
PY
find $OUT -name '*.csv' -size +1M -delete
