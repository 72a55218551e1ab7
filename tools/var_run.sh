#!/bin/bash
# GPU box: bench each era-zkevm_circuits_amd/libzkgl_var_<TAG>.so (tools/variants.sh) next to the real library.  Stubbed variants give WRONG values.
cd "$(dirname "$0")/.."
B=${B:-64}
one() {
  ZKGL_STUB_RUN=1 ZKGL_LIB=$2 timeout 600 python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline 2>gpurun_out/var_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s seed %.3f s step %.2f ms  k_witness_loop %.2f ms gates %.2f outer %.2f' % ('$1', d['config']['input_seeding_s'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_kernels_ms']['k_check_gates_loop'], d['roofline']['other_kernels_ms']['outer_post_and_checks_overlapped']))" || tail -3 gpurun_out/var_err.txt
}
one full "$(pwd)/era-zkevm_circuits_amd/libzkgl.so"
for f in era-zkevm_circuits_amd/libzkgl_var_*.so; do t=${f##*_var_}; one "${t%.so}" "$(pwd)/$f"; done
