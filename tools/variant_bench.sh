#!/bin/bash
# usage (GPU box, repo root): tools/variant_bench.sh libA.so libB.so ...   (variants built with ZKGL_OUT/ZKGL_DEFS)
# prints k_witness_loop time and dynamic instruction counts per variant (B=145).
for lib in "$@"; do
  export ZKGL_LIB=$(pwd)/era-zkevm_circuits_amd/$lib
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline < /dev/null 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'step', round(d['ms_per_step'],2), 'k_witness_loop', round(d['roofline']['avg_launch_ms'],2), 'checksum', d['commitment_checksum'])"
done
