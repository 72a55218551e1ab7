#!/bin/bash
# GPU box, repo root: A/B of k_witness_loop variants (tools/variants.sh builds them) on the bench command -> gpurun_out/r4_loop_variants.txt
# usage: tools/loop_variants_r4.sh [TAG ...]   (default: the product library + every libzkgl_var_*.so present)
set -u
ROOT=$(pwd); mkdir -p gpurun_out
OUT=gpurun_out/r4_loop_variants.txt
: > $OUT
run() {  # $1 label, $2 lib ("" = product), rest: bench args
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export ZKGL_LIB=$lib; else unset ZKGL_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" 2> /tmp/var_$label.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', 'value %.1f G' % (d['value']/1e9), 'step %.2f ms' % d['ms_per_step'], 'loop %.2f ms' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'clk %.0f' % r['shader_clock_mhz'], 'skipped %.3f' % r.get('witness_only_permutations_skipped_frac', -1), 'resident %.1f G' % (d['value_inputs_resident']/1e9))
" >> $OUT 2>&1 || { echo "$label FAILED" >> $OUT; tail -3 /tmp/var_$label.err >> $OUT; }
}
run product "" 
run product_realistic "" --fixture realistic
for so in era-zkevm_circuits_amd/libzkgl_var_*.so; do
  tag=$(basename $so .so); tag=${tag#libzkgl_var_}
  case " ${SKIP_TAGS:-} " in *" $tag "*) continue;; esac
  run $tag $ROOT/$so
done
run product_again ""
unset ZKGL_LIB
cat $OUT
