for t in INV FIND FMA P2 SL ALLV; do
  ZKGL_STUB_RUN=1 ZKGL_LIB=$PWD/era-zkevm_circuits_amd/libzkgl_var_$t.so CONFIGS=C3k timeout 300 python tests/config_timings.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$t', 'step_ms', d['step_ms'], 'loop_ms', d['k_witness_loop_ms'])"
done
