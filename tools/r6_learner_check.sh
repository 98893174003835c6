#!/bin/bash
# Round 6: the persistent learner after the fault's root cause - GPU tests of both exchange modes, the exact gather test, the assertion build's
# whole-update soak, and the step time in both modes.  gpurun -- bash tools/r6_learner_check.sh
O=gpurun_out/r6_learner
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_learner.py -m gpu -x -q -k "persistent" > $O/pytest_persistent.log 2>&1
tail -5 $O/pytest_persistent.log
for m in auto agent; do MODE=$m timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > $O/time_$m.json; done
PROF=0 timeout 300 python tools/time_learner_persistent.py 2>&1 | tail -1 > $O/time_prof0.json
python - <<'PY'
import json
for m in ("auto","agent","prof0"):
    try:
        d=json.load(open(f"gpurun_out/r6_learner/time_{m}.json"))
        print(m, {k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if "us_per_step" in k or k=="status"})
    except Exception as e: print(m, "ERR", e)
PY
