#!/bin/bash
# Round 6, re-entry check (one gpurun call): the library rebuilt from the committed sources in a re-created container (same build id / lib sha16 as the
# round's evidence) runs the driver's own sequence on a fresh lease: GPU tests, smoke(), the driver's bench command, the no-flags bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_reentry
mkdir -p $O
export Q1_TUNABLEOP=0
python -c "import q1physrl_amd._lib as L, q1physrl_amd.build as B; print('build id', B.sources_sha16(), 'lib sha16', L.lib_sha16())" > $O/build_id.txt 2>&1
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 ) > $O/pytest_gpu.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20.json 2> $O/bench_driver.err
( time timeout 900 python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err ) 2> $O/bench_noflags.time
cat $O/build_id.txt $O/pytest_gpu.txt $O/smoke.txt; cut -c1-700 $O/bench_driver_steps20.json; cut -c1-300 $O/bench_noflags.json; tail -3 $O/bench_noflags.time
