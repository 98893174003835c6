#!/bin/bash
# round 4, final pass on the GPU box: the evidence set of the tree as it is + the bench records kept under profiles/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WHAT=${1:-all}
bash tools/profile_round.sh r4 $WHAT > gpurun_out/prof_r4.log 2>&1
O=gpurun_out/r4_final
mkdir -p $O
# profiles/pmc.json of THIS tree first, so that the bench lines below carry measured traffic (roofline.traffic) next to the algorithmic bytes
cp gpurun_out/prof_r4/pmc.json profiles/pmc.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20.json 2> $O/bench_driver_steps20.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --envs 131072 --no-cpu-baseline --no-secondary > $O/bench_131072.json 2> $O/bench_131072.err
timeout 600 python bench.py --envs 262144 --config params_yml --no-secondary > $O/bench_configs2_262144.json 2> $O/bench_configs2.err
Q1_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_1gpu.json 2> $O/bench_2rank_1gpu.err
Q1_BENCH_OVERSUBSCRIBE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2rank_1gpu_torchrun.json 2> $O/bench_2rank_1gpu_torchrun.err
timeout 900 python tools/bench_configs.py > $O/bench_configs.txt 2> $O/bench_configs.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_final/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f.split("/")[-1], "n_gpus", d["n_gpus"], "value %.2f G"%(d["value"]/1e9), "ms/step %.5f"%d["ms_per_step"], "w/e %.3f"%r["wall_over_event"], "launch_us %.2f"%r["avg_launch_us"], "frac %.3f"%r["frac"], "traffic", r["traffic"], "stale", r["pmc_stale"])
    except Exception as ex:
        print(f, "ERR", ex)
P
cat $O/bench_configs.txt | cut -c1-200; cat $O/smoke.txt | tail -2
