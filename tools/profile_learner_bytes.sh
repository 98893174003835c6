#!/bin/bash
# On the GPU box: HBM bytes of ONE native SGD step (32 768 samples, tools/time_learner.py --phase step), per kernel: rocprofv3
# --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md: KiB units, FETCH_SIZE x 2 on gfx950,
# calibrated on calib_copy_kernel in tools/profile_round.sh).  VERDICT r3 item 4 asked for bytes per step next to time.
set -u
TAG=${1:-r4}
MODE=${2:-auto}          # q1env_learner_sgd_step's kernel sequence: auto | four_launch | fused | fused_dw1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_learner_bytes_${TAG}_$MODE
mkdir -p $OUT
CMD="python tools/time_learner.py --phase step --steps 40 --step-mode $MODE"
$CMD > $OUT/unprofiled.json 2> $OUT/unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > /dev/null 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o t -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o t -- $CMD > /dev/null 2> $OUT/write.err
python - <<PY
import csv, glob, collections
out = "$OUT"
print(open(out + "/unprofiled.json").read().strip())
t = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0].replace("q1learn::", "").replace("void ", "")
        if "learner" in k or "ppo_loss" in k or "adam" in k:
            t[k] = float(r["AverageNs"]) / 1e3
b = collections.defaultdict(lambda: collections.defaultdict(list))
for p, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("q1learn::", "").replace("void ", "")
            if r["Counter_Name"] == c and ("learner" in k or "ppo_loss" in k or "adam" in k):
                b[k][c].append(float(r["Counter_Value"]) * 1024.0)
tot_t = tot_b = 0.0
print(f"{'kernel':34s} {'us':>8s} {'fetch x2 MB':>12s} {'write MB':>10s} {'TB/s':>6s}")
for k in sorted(t, key=lambda x: -t[x]):
    fx = 2.0 * sum(b[k]["FETCH_SIZE"]) / max(1, len(b[k]["FETCH_SIZE"]))
    wr = sum(b[k]["WRITE_SIZE"]) / max(1, len(b[k]["WRITE_SIZE"]))
    tot_t += t[k]; tot_b += fx + wr
    print(f"{k[:34]:34s} {t[k]:8.2f} {fx / 1e6:12.2f} {wr / 1e6:10.2f} {(fx + wr) / t[k] / 1e6:6.2f}")
print(f"{'sum of the step kernels':34s} {tot_t:8.2f} {'':12s} {tot_b / 1e6:10.1f} MB per step in all")
PY
find $OUT -name '*.csv' -size +1M -delete
