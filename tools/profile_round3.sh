#!/bin/bash
# Run on the GPU box (through gpurun): for every bench.py mode (rollout = the default, step, server) at 65 536 envs, and for the
# per-tick step kernel at 1 M and 4 M envs: rocprofv3 --kernel-trace --stats, then SEPARATE PMC passes of the SAME command
#   FETCH_SIZE | WRITE_SIZE | SQ issue counters | SQ instruction-type counters | GRBM activity
# (one counter group per pass, never combined with a trace domain other than --kernel-trace), then the known-bytes calibration kernel.
# tools/summarize_pmc.py turns the CSVs into gpurun_out/prof_<tag>/{summary.txt,pmc.json}; copy those into profiles/.
set -u
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-secondary --steps 1440 --warmup 720"
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
SQ3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY"
run_set() {   # name, bench args, which passes ("all" or "bytes")
    local name=$1 args=$2 what=$3
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -o t -- python bench.py $args $COMMON > $OUT/$name.bench_trace.json 2> $OUT/$name.trace.err
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$name/fetch -o t -- python bench.py $args $COMMON > /dev/null 2> $OUT/$name.fetch.err
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$name/write -o t -- python bench.py $args $COMMON > /dev/null 2> $OUT/$name.write.err
    if [ "$what" = "all" ]; then
        rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/$name/sq1 -o t -- python bench.py $args $COMMON > /dev/null 2> $OUT/$name.sq1.err
        rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/$name/sq2 -o t -- python bench.py $args $COMMON > /dev/null 2> $OUT/$name.sq2.err
        rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $OUT/$name/sq3 -o t -- python bench.py $args $COMMON > /dev/null 2> $OUT/$name.sq3.err
        rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$name/grbm -o t -- python bench.py $args $COMMON > /dev/null 2> $OUT/$name.grbm.err
    fi
    python bench.py $args $COMMON > $OUT/$name.bench_unprofiled.json 2> /dev/null     # the same command without the profiler (event time)
}
run_set rollout_65536 "--mode rollout" all
run_set step_65536 "--mode step" all
run_set server_65536 "--mode server" all
if [ "${2:-}" != "small" ]; then
    run_set step_262144 "--mode step --envs 262144" bytes
    run_set step_1048576 "--mode step --envs 1048576" bytes
    run_set step_4194304 "--mode step --envs 4194304 --steps 288 --warmup 72" bytes
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib/fetch -o t -- python tools/calib_traffic.py > /dev/null 2> $OUT/cal_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib/write -o t -- python tools/calib_traffic.py > /dev/null 2> $OUT/cal_write.err
python tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.csv' -size +1M -delete
