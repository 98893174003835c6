#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the bench command.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r1}
shift || true
ARGS=${*:-"--no-cpu-baseline --steps 1440 --warmup 720"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name '*.csv' | head -20
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files
find $OUT -name '*kernel_trace.csv' -size +3M -delete
find $OUT -name '*counter_collection.csv' -size +3M -delete
