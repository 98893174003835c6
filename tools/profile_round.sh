#!/bin/bash
# Run on the GPU box (through gpurun): the round's evidence set for EVERY BASELINE config from the tree as it is.
#   rollout_65536          configs[1]: bench.py --mode rollout                        (rollout_kernel<float, true, 2, false, 1, false, 2>: 720-tick launches)
#   rollout_131072         configs[3] shard: --mode rollout --envs 131072             (same instantiation, two waves per SIMD)
#   rollout_params_262144  configs[2]: --config params_yml --envs 262144, in-kernel reset (rollout_kernel<float, true, 2, true, 1, false, 2>)
#   step_65536 / step_262144 / step_1048576 / step_4194304   the per-tick kernel (HBM-bound formulation), server_65536 the LDS pair
#   sampler_32768 / sampler_262144   configs[4]: kernel-trace statistics of the sampler loop (mlp_forward_kernel, sample / step / resident)
# For each bench set: rocprofv3 --kernel-trace --stats, then SEPARATE PMC passes of the SAME command
#   FETCH_SIZE | WRITE_SIZE | SQ issue counters | SQ instruction-type counters | GRBM activity
# (one counter group per pass, never combined with a trace domain other than --kernel-trace), then the known-bytes calibration kernel.
# tools/summarize_pmc.py turns the CSVs into gpurun_out/prof_<tag>/{summary.txt,pmc.json} (each entry carries the build id of the
# library it profiled); copy those into profiles/.
set -u
TAG=${1:-r5}
WHAT=${2:-all}
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
    for f in $OUT/$name/trace/*/*kernel_stats.csv $OUT/$name/trace/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv; done
}
run_set rollout_65536 "--mode rollout" all
run_set rollout_131072 "--mode rollout --envs 131072" all
run_set rollout_params_262144 "--mode rollout --config params_yml --envs 262144" all
if [ "$WHAT" != "rollout" ]; then
    run_set step_65536 "--mode step" all
    run_set server_65536 "--mode server" all
    run_set step_262144 "--mode step --envs 262144" bytes
    run_set step_1048576 "--mode step --envs 1048576" bytes
    run_set step_4194304 "--mode step --envs 4194304 --steps 288 --warmup 72" bytes
    # configs[4]: the sampler loop with the policy forward in it (two-launch tick and resident sampler), kernel-trace statistics
    for n in 32768 262144; do
        rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sampler_$n/trace -o t -- python tools/profile_sampler.py $n > $OUT/sampler_$n.txt 2> $OUT/sampler_$n.err
        for f in $OUT/sampler_$n/trace/*/*kernel_stats.csv $OUT/sampler_$n/trace/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" $OUT/sampler_${n}_kernel_stats.csv; done
    done
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib/fetch -o t -- python tools/calib_traffic.py > /dev/null 2> $OUT/cal_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib/write -o t -- python tools/calib_traffic.py > /dev/null 2> $OUT/cal_write.err
python tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.csv' -size +1M -delete
