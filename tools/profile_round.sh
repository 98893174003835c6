#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE) for the
# bench command and for the traffic-calibration kernel.  Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you
# want judged into profiles/.
set -u
TAG=${1:-r2}
shift || true
ARGS=${*:-"--no-cpu-baseline --no-secondary --steps 1440 --warmup 720"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -o fetch -- python tools/calib_traffic.py > /dev/null 2> $OUT/cal_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -o write -- python tools/calib_traffic.py > /dev/null 2> $OUT/cal_write.err
# the per-tick kernels (bench.py's default mode is the resident tick server)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_step -o trace -- python bench.py --mode step $ARGS > $OUT/bench_step_trace.json 2> $OUT/trace_step.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_step -o fetch -- python bench.py --mode step $ARGS > /dev/null 2> $OUT/fetch_step.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_step -o write -- python bench.py --mode step $ARGS > /dev/null 2> $OUT/write_step.err
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.csv' -size +2M -delete
