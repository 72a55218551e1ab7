#!/bin/bash
# tools/emulated_race_check.sh [pytest node ids] — -m gpu tests on the EMULATED DEVICE under the DATA-RACE DETECTOR (tests/emu/README.md): the
# kernels' translation unit compiled with -fsanitize=thread, every work-item a ThreadSanitizer fiber; the only happens-before edges are the
# device's (wavefront operations among the lanes that meet, __syncthreads in the workgroup, launch after launch).  Reports: $OUT/report.<pid>
# (default OUT=/tmp/emulated_race), pytest's output: $OUT/pytest.log; the last lines count the reports.  EMU_TSAN_MODE=grid: the second pass (races BETWEEN the workgroups of a launch; LDS exempt).  5-10x slower than the plain emulated device: pick the tests.
set -uo pipefail
cd "$(dirname "$0")/.."
# extra defines under the detector: EMU_VARIANT=tsan_probe DEFS="-DZKGL_EXPERIMENT=2" tools/emulated_race_check.sh ...
V=${EMU_VARIANT:-tsan}
EMU_TSAN=1 EMU_VARIANT=$V bash tests/emu/dev/build.sh ${DEFS:-} | tail -1 || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
OUT=${OUT:-/tmp/emulated_race}; mkdir -p $OUT; rm -f $OUT/report.* $OUT/pytest.log
LD_PRELOAD=$RT TSAN_OPTIONS="halt_on_error=0 log_path=$OUT/report report_signal_unsafe=0" ZKGL_LIB=$PWD/tests/emu/_gen/dev_$V/libzkgl.so \
  python -m pytest -m gpu -q -p no:cacheprovider -n ${JOBS:-6} --timeout 6000 "$@" 2>&1 | tee $OUT/pytest.log | tail -3
echo "data races reported: $(cat $OUT/report.* 2>/dev/null | grep -c 'WARNING: ThreadSanitizer')"
cat $OUT/report.* 2>/dev/null | grep SUMMARY | sed 's/ (lib.*//; s#.*/src/##' | sort | uniq -c | sort -rn | head -40
exit 0
