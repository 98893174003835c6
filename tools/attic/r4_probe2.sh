#!/bin/bash
# round 4, probe 2: whole GPU suite on the slimmed tick + driver-style bench lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_probe2
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for rep in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_20_$rep.json 2> $O/bench_20_$rep.err
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_probe2/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]; hs=r["host_split_us"]
        print(f.split("/")[-1], "value %.2f G"%(d["value"]/1e9), "ms/step %.4f us"%(d["ms_per_step"]*1e3), "w/e %.3f"%r["wall_over_event"], "launch_us %.2f"%r["avg_launch_us"], "frac %.3f"%r["frac"], {k:(round(v,2) if isinstance(v,float) else v) for k,v in hs.items()})
        ss=d.get("steady_state_720_ticks")
        if ss: print("   steady:", {k:(round(v.get("us_per_tick",0),3), round(v.get("env_steps_per_s",0)/1e9,1)) for k,v in ss.items()})
        cb=d.get("cpu_baseline")
        if cb: print("   cpu:", cb.get("value"), cb.get("parity_vs_gpu_after_719_ticks"))
    except Exception as ex:
        print(f, "ERR", ex)
P
tail -15 $O/pytest.log
