#!/bin/bash
# tools/emulated_bounds_check.sh [pytest node ids] — -m gpu tests on the EMULATED DEVICE under the ADDRESS SANITIZER (tests/emu/README.md): the
# kernels' translation unit compiled with -fsanitize=address; every hipMalloc is a heap block of its exact size, LDS arrays are globals with
# red zones.  The kernels address the store through buffer descriptors without bounds (num_records = -1): an access past a buffer returns
# garbage silently on the device — here it stops the run with both stacks.  Reports: $OUT/report.<pid>, pytest's output $OUT/pytest.log.
set -uo pipefail
cd "$(dirname "$0")/.."
# extra defines (e.g. an elimination-probe build): EMU_VARIANT=asan_probe DEFS="-DZKGL_EXPERIMENT=2" tools/emulated_bounds_check.sh ...
V=${EMU_VARIANT:-asan}
EMU_ASAN=1 EMU_VARIANT=$V bash tests/emu/dev/build.sh ${DEFS:-} | tail -1 || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
OUT=${OUT:-/tmp/emulated_bounds}; mkdir -p $OUT; rm -f $OUT/report.* $OUT/pytest.log
LD_PRELOAD=$RT ASAN_OPTIONS="detect_leaks=0 halt_on_error=0 log_path=$OUT/report detect_stack_use_after_return=0 allocator_may_return_null=1" ZKGL_LIB=$PWD/tests/emu/_gen/dev_$V/libzkgl.so \
  python -m pytest -m gpu -q -p no:cacheprovider -n ${JOBS:-6} --timeout 6000 "$@" 2>&1 | tee $OUT/pytest.log | tail -3
echo "address errors reported: $(cat $OUT/report.* 2>/dev/null | grep -c 'ERROR: AddressSanitizer')"
cat $OUT/report.* 2>/dev/null | grep -A3 "ERROR: AddressSanitizer" | head -40
exit 0
