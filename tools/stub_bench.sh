#!/bin/bash
# Time attribution of k_witness_loop by elimination (results are WRONG in every stubbed variant: never ship them).
#   tools/stub_bench.sh build   (container: builds era-zkevm_circuits_amd/libzkgl_stub_<tag>.so for each combination of
#                                ZKGL_STUB_STORES / ZKGL_STUB_LOADS / ZKGL_STUB_P2 in csrc/kernels_engine.hpp)
#   tools/stub_bench.sh run     (GPU box, repo root: bench each variant at B=145, print the kernel's average launch time)
set -euo pipefail
cd "$(dirname "$0")/.."
declare -A V=( [S]="-DZKGL_STUB_STORES" [L]="-DZKGL_STUB_LOADS" [P]="-DZKGL_STUB_P2"
               [SL]="-DZKGL_STUB_STORES -DZKGL_STUB_LOADS" [SP]="-DZKGL_STUB_STORES -DZKGL_STUB_P2"
               [LP]="-DZKGL_STUB_LOADS -DZKGL_STUB_P2" [SLP]="-DZKGL_STUB_STORES -DZKGL_STUB_LOADS -DZKGL_STUB_P2" )
ORDER="S L P SL SP LP SLP"
if [ "${1:-run}" = build ]; then
  for t in $ORDER; do ZKGL_OUT=../libzkgl_stub_$t.so ZKGL_DEFS="${V[$t]}" era-zkevm_circuits_amd/build.sh; done
  era-zkevm_circuits_amd/build.sh   # leave build/ holding the objects of the real library
  exit 0
fi
one() {  # $1 = label, $2 = library
  ZKGL_LIB=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline < /dev/null 2>gpurun_out/stub_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-5s step %.2f ms  k_witness_loop %.2f ms' % ('$1', d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
one full "$(pwd)/era-zkevm_circuits_amd/libzkgl.so"
export ZKGL_STUB_RUN=1
for t in $ORDER; do one "-$t" "$(pwd)/era-zkevm_circuits_amd/libzkgl_stub_$t.so"; done
