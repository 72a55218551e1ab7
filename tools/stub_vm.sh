#!/bin/bash
# Time attribution of the main_vm witness kernel by elimination (stubbed variants give WRONG results: never ship them) + a kernel trace
# of the real step.  Build the variants first (tools/stub_bench.sh build, or the S / L / SL / SLP subset).
cd "$(dirname "$0")/.."
B=${B:-64}
one() {
  ZKGL_STUB_RUN=1 ZKGL_LIB=$2 timeout 600 python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline 2>gpurun_out/stub_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-5s step %.2f ms  k_witness_loop %.2f ms gates %.2f outer %.2f' % ('$1', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_kernels_ms']['k_check_gates_loop'], d['roofline']['other_kernels_ms']['outer_post_and_checks_overlapped']))"
}
one full "$(pwd)/era-zkevm_circuits_amd/libzkgl.so"
for t in S L SL SLP; do [ -f era-zkevm_circuits_amd/libzkgl_stub_$t.so ] && one "-$t" "$(pwd)/era-zkevm_circuits_amd/libzkgl_stub_$t.so"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_vm -o vm -- python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline > gpurun_out/prof_vm_bench.json 2>gpurun_out/prof_vm_err.txt
ls gpurun_out/prof_vm | head; find gpurun_out/prof_vm -name "*kernel_stats*" | head -3 | while read f; do head -25 "$f"; done
