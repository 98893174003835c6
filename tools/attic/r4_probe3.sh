#!/bin/bash
# round 4, probe 3: whole GPU suite; streaming step kernel A/B; tanh micro-benchmark; stream-priority A/B of the driver-style line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_probe3
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 900 python tools/exp_step_large.py run > $O/exp_step_large.txt 2>&1
timeout 120 ./gpurun_scratch/ubench_tanh > $O/ubench_tanh.txt 2>&1
timeout 900 python tools/soak_oracle.py --rounds 24 > $O/soak_oracle.txt 2>&1
timeout 600 python tools/soak_pair.py > $O/soak_pair.txt 2>&1
timeout 600 python tools/soak_resident.py > $O/soak_resident.txt 2>&1
for rep in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_20_$rep.json 2> $O/bench_20_$rep.err
  Q1ENV_STREAM_PRIORITY=high timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_20_prio_$rep.json 2> $O/bench_20_prio_$rep.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_probe3/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]; hs=r["host_split_us"]
        print(f.split("/")[-1], "value %.2f G"%(d["value"]/1e9), "w/e %.3f"%r["wall_over_event"], "launch_us %.2f"%r["avg_launch_us"], "frac %.3f"%r["frac"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in hs.items()})
    except Exception as ex:
        print(f, "ERR", ex)
P
tail -2 $O/soak_oracle.txt; tail -2 $O/soak_pair.txt; tail -2 $O/soak_resident.txt; cat $O/exp_step_large.txt | cut -c1-330; cat $O/ubench_tanh.txt; tail -8 $O/pytest.log
