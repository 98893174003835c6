#!/bin/bash
O=gpurun_out/r6_f32
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_learner.py -m gpu -x -q -s -k "f32 or gathers" > $O/pytest_f32.log 2>&1
tail -15 $O/pytest_f32.log
for m in auto agent; do MODE=$m timeout 300 python tools/time_learner_f32.py 2>&1 | tail -1 > $O/time_f32_$m.json; cat $O/time_f32_$m.json; done
