#!/bin/bash
# one development iteration of the persistent learner on the GPU box: equivalence tests, then the timing tool without and with phase stamps
# (PROF=<g>: wave 0 of workgroup g of the policy group is stamped); PROF_GS="0 3 7" stamps several workgroups in turn
mkdir -p gpurun_out/pl
timeout 600 python -m pytest tests/test_hip_learner.py -q -x -k "persistent" 2>&1 | tail -3
timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > gpurun_out/pl/time.json
for g in ${PROF_GS:-0}; do PROF=$g timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > gpurun_out/pl/time_prof$g.json; done
python - <<'PY'
import json,glob
for f in ["gpurun_out/pl/time.json"]+sorted(glob.glob("gpurun_out/pl/time_prof*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); print(open(f).read()[-2000:]); continue
    print(f.split("/")[-1], [round(d[f"persistent_rep{r}_us_per_step"],2) for r in range(3)], "status", d.get("status"))
    if "prof_us_per_step" in d: print({k: round(v,2) for k,v in d["prof_us_per_step"].items() if k!="-"})
PY
