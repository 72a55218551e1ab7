#!/bin/bash
# tools/emulated_variants.sh — the PARITY half of tools/ab_r5.sh on the EMULATED DEVICE (tests/emu/README.md): every opt-in library of
# tools/variants_r5.sh built for the host with the same switches, the same tests.  No GPU, no timing.  ~25 min on 8 cores.  Log: $OUT (default
# /tmp/emulated_variants.log); one line per group: [name] <pytest summary>
set -uo pipefail
cd "$(dirname "$0")/.."
OUT=${OUT:-/tmp/emulated_variants.log}; : > $OUT
G=$PWD/tests/emu/_gen
b() { EMU_VARIANT=$1 bash tests/emu/dev/build.sh $2 2>&1 | grep -E "error|failed" ; }
b p2m_binv "-DZKGL_P2_MERGE -DZKGL_BATCH_INV"; b p2m "-DZKGL_P2_MERGE"; b binv "-DZKGL_BATCH_INV"; b chains "-DZKGL_SELECT_CHAINS_KERNEL"
b k8 "-DZKGL_BYTEBUF_KERNEL -DZKGL_STRAND_PLANES_KERNEL"; b sha4 "-DZKGL_SHA4_KERNEL"
t() {  # name, environment..., -- pytest arguments
  local name=$1; shift; local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local r; r=$(env "${envs[@]}" python -m pytest -m gpu -q -n ${JOBS:-7} -p no:cacheprovider --timeout 3000 "$@" 2>&1 | tail -1)
  echo "[$name] $r" | tee -a $OUT
}
VM="tests/test_gpu_main_vm.py tests/test_gpu_store_tiling.py::test_main_vm_under_a_forced_tiling tests/test_gpu_cs.py::test_vm_shaped_gpu_equals_oracle tests/test_gpu_cs.py::test_ram_fixture_trace_bit_exact tests/test_gpu_cs.py::test_storage_validity_gpu_equals_oracle tests/test_fused_check.py tests/test_fuzz_programs.py"
for v in p2m_binv p2m binv; do t $v ZKGL_LIB=$G/dev_$v/libzkgl.so -- $VM; done
t chains ZKGL_LIB=$G/dev_chains/libzkgl.so ZKGL_SELECT_CHAINS=1 -- $VM
t bytebuf ZKGL_LIB=$G/dev_k8/libzkgl.so -- tests/test_zz_round5_gpu.py -k bytebuf
t strand_planes ZKGL_LIB=$G/dev_k8/libzkgl.so ZKGL_STRAND_PLANES=1 ZKGL_STRANDS=1 -- tests/test_gpu_cs.py tests/test_fuzz_programs.py tests/test_fused_check.py --deselect tests/test_gpu_cs.py::test_linear_hasher_gpu --deselect tests/test_gpu_cs.py::test_narrow_strand_form_gpu
t bytebuf_and_planes ZKGL_LIB=$G/dev_k8/libzkgl.so ZKGL_STRAND_PLANES=1 ZKGL_BYTEBUF_MACRO=1 -- tests/test_zz_round5_gpu.py -k bytebuf
t sha4 ZKGL_LIB=$G/dev_sha4/libzkgl.so -- tests/test_zz_round5_gpu.py -k sha4
t sha4_tables ZKGL_LIB=$G/dev_sha4/libzkgl.so ZKGL_SHA4_MACRO=1 -- tests/test_sha256_reference_tables.py
t sha4_forged ZKGL_LIB=$G/dev_sha4/libzkgl.so ZKGL_SHA4_MACRO=1 -- tests/test_zz_round5_gpu.py -k forged
