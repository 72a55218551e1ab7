#!/bin/bash
# GPU box: strands per tile (ZKGL_STRANDS_PER_TILE, a build-time constant of host and device: variants built with
# ZKGL_DEFS=-DZKGL_STRANDS_PER_TILE=n ZKGL_OUT=../libzkgl_var_STn.so ZKGL_BUILD_DIR=/tmp/build_stn era-zkevm_circuits_amd/build.sh)
# over the configurations whose loop or outer scopes run the strand kernels -> gpurun_out/strands_ab.txt
ROOT=$(pwd); : > gpurun_out/strands_ab.txt
for t in ${VARIANTS:-16 8 4}; do
  lib=$ROOT/era-zkevm_circuits_amd/libzkgl_var_ST$t.so; [ $t = 16 ] && lib=$ROOT/era-zkevm_circuits_amd/libzkgl.so
  [ -f $lib ] || continue
  for c in ${CFGS:-C3k C3s C5 C1 C4s}; do
    ZKGL_LIB=$lib CONFIGS=$c timeout 600 python tests/config_timings.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('strands=$t', d['config'][:60], 'step_ms', d['step_ms'], 'loop_ms', d['k_witness_loop_ms'], 'seed_s', d.get('seed_s'), 'seeded_ok', d.get('seeded_equals_native'))" >> gpurun_out/strands_ab.txt
  done
done
cat gpurun_out/strands_ab.txt
