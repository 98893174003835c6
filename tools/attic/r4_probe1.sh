#!/bin/bash
# round 4, probe 1: the completion-signal path on hardware + A/B of the runtime knobs that touch launch / completion latency
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_probe1
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_signal.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for rep in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_signal_$rep.json 2> $O/bench_signal_$rep.err
done
Q1_BENCH_NO_SIGNAL=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_nosignal.json 2> $O/bench_nosignal.err
HSA_ENABLE_INTERRUPT=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_signal_nointr.json 2> $O/bench_signal_nointr.err
Q1_BENCH_SIGNAL_MARK=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_signalmark.json 2> $O/bench_signalmark.err
HIP_FORCE_DEV_KERNARG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_signal_devkernarg.json 2> $O/bench_signal_devkernarg.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_probe1/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]; hs=r["host_split_us"]
        print(f.split("/")[-1], "value %.2f G"%(d["value"]/1e9), "ms/step %.4f us"%(d["ms_per_step"]*1e3), "w/e %.3f"%r["wall_over_event"], "launch_us %.2f"%r["avg_launch_us"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in hs.items()})
    except Exception as ex:
        print(f, "ERR", ex)
P
tail -5 $O/pytest.log
