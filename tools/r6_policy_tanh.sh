#!/bin/bash
# Round 6, VERDICT r5 item 5: the policy forward with the activation's tail in packed float16 (-DQ1POL_ACT_PK16, libq1env_pk16.so) against the
# product library: kernel time at 262 144 / 32 768 rows, error against the float32 torch modules, the resident sampler's tick.
O=gpurun_out/r6_tanh
mkdir -p $O
[ -f q1physrl_amd/libq1env_pk16.so ] || python -c "
from q1physrl_amd import build; import os
build.build_lib(extra_flags=['-DQ1POL_ACT_PK16'], out=os.path.join(build.PKG, 'libq1env_pk16.so'), tag='_pk16')"
for lib in "" q1physrl_amd/libq1env_pk16.so; do
  tag=${lib:+pk16}; tag=${tag:-product}
  env ${lib:+Q1ENV_LIB_PATH=$PWD/$lib} timeout 300 python tools/time_mlp.py 32768 262144 > $O/time_mlp_$tag.txt 2>&1
  env ${lib:+Q1ENV_LIB_PATH=$PWD/$lib} timeout 300 python tools/mlp_error.py > $O/mlp_error_$tag.txt 2>&1
  env ${lib:+Q1ENV_LIB_PATH=$PWD/$lib} timeout 300 python tools/bench_resident.py > $O/resident_$tag.txt 2>&1
  echo "== $tag"; tail -2 $O/time_mlp_$tag.txt; tail -2 $O/mlp_error_$tag.txt; tail -3 $O/resident_$tag.txt | cut -c1-400
done
