#!/bin/bash
# tests/emu/dev/build.sh [defs...] — build the product for the EMULATED DEVICE (tests/emu/README.md): every source of era-zkevm_circuits_amd/csrc,
# re-written by gen_dev.py, compiled as host C++ over tests/emu/dev/hip/hip_runtime.h, linked with the fiber scheduler (emu_rt.cpp).
#   -> tests/emu/_gen/dev/libzkgl.so (+ libzkgl_testcircuits.so beside it): load it with ZKGL_LIB=<that path>.  TEST INFRASTRUCTURE.
# EMU_VARIANT=<name> with defs (e.g. EMU_VARIANT=probe build.sh -DZKGL_EXPERIMENT=2): the device source with extra defines -> tests/emu/_gen/dev_<name>/
# EMU_TSAN=1: the data-race detector build (emu_rt.cpp: every work-item a ThreadSanitizer fiber) -> tests/emu/_gen/dev_tsan/; run under
#   LD_PRELOAD=/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so  (tools/emulated_race_check.sh)
# EMU_ASAN=1: the address-sanitizer build (out-of-bounds / use-after-free in kernels: device buffers, LDS, local arrays) -> tests/emu/_gen/dev_asan/;
#   run under LD_PRELOAD=.../libclang_rt.asan-x86_64.so  (tools/emulated_bounds_check.sh)
# EMU_OPT=-O2: the kernels' translation unit at -O2 (runs twice as fast, compiles in ~100 s instead of ~10) -> tests/emu/_gen/dev_O2/
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../../.." && pwd)
OPT=${EMU_OPT:--O1}
TSAN_CC=""; TSAN_LD=""
if [ -n "${EMU_ASAN:-}" ]; then EMU_VARIANT=${EMU_VARIANT:-asan}; set -- "$@" -DEMU_ASAN=1; TSAN_CC="-fsanitize=address"; TSAN_LD="-fsanitize=address -shared-libsan"; fi
if [ -n "${EMU_TSAN:-}" ]; then EMU_VARIANT=${EMU_VARIANT:-tsan}; set -- "$@" -DEMU_TSAN=1; TSAN_CC="-fsanitize=thread"; TSAN_LD="-fsanitize=thread -shared-libsan"; fi
# EMU_ASAN=1 EMU_SAN_HOST=1: the HOST sources (recorder, packers, C ABI) under the address sanitizer too -> tests/emu/_gen/dev_asanhost/ (the `-m "not gpu"` tests through ZKGL_LIB)
HOST_SAN=""; if [ -n "${EMU_SAN_HOST:-}" ] && [ -n "${EMU_ASAN:-}" ]; then HOST_SAN="$TSAN_CC"; [ "${EMU_VARIANT:-asan}" != asan ] || EMU_VARIANT=asanhost; fi
if [ -n "${EMU_SAN_HOST:-}" ] && [ -n "${EMU_TSAN:-}" ]; then HOST_SAN="$TSAN_CC"; [ "${EMU_VARIANT:-tsan}" != tsan ] || EMU_VARIANT=tsanhost; fi   # the host pool's OS threads under the race detector
# EMU_UBSAN=1: the host sources under the undefined-behaviour sanitizer (shifts, signed overflow, misaligned or null accesses, bad enum / bool loads) -> tests/emu/_gen/dev_ubsan/;
#   run under LD_PRELOAD=.../libclang_rt.ubsan_standalone-x86_64.so with UBSAN_OPTIONS=print_stacktrace=1
if [ -n "${EMU_UBSAN:-}" ]; then EMU_VARIANT=${EMU_VARIANT:-ubsan}; HOST_SAN="-fsanitize=undefined -fno-sanitize=vptr,function"; TSAN_LD="-fsanitize=undefined -shared-libsan"; fi
GEN=$HERE/../_gen/dev${EMU_VARIANT:+_$EMU_VARIANT}; [ "$OPT" = "-O1" ] || GEN=${GEN}_${OPT#-}
mkdir -p $GEN/obj $GEN/obj/testing
python $HERE/gen_dev.py $GEN
CXX=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-std=c++17 -O1 -g1 -fPIC -fno-omit-frame-pointer -Wno-unknown-attributes -Wno-ignored-attributes -Wno-macro-redefined -Wno-unused-value -Wno-pass-failed -Wno-keyword-macro -Wno-deprecated-declarations -I$HERE -I$GEN/src $*"
if [ "$(cat $GEN/.flags 2>/dev/null)" != "$FLAGS" ]; then rm -f $GEN/obj/*.o $GEN/obj/testing/*.o; echo "$FLAGS" > $GEN/.flags; fi
NEWEST_HDR=$(ls -t $GEN/src/*.hpp $GEN/src/circuits/*.hpp $ROOT/include/*.h $HERE/hip/*.h $HERE/rccl/*.h | head -1)
pids=(); fails=0
cc() {  # $1 source, $2 object, $3 extra flags
  if [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ "$NEWEST_HDR" -nt "$2" ]; then $CXX $FLAGS ${3:-} -c "$1" -o "$2" & pids+=($!); fi
}
for f in $GEN/src/*.cpp $GEN/src/circuits/*.cpp; do
  if [ $(basename $f) = zkgl_device.cpp ]; then cc $f $GEN/obj/$(basename $f).o "$OPT $TSAN_CC ${EMU_UBSAN:+$HOST_SAN}"   # the kernels
  else cc $f $GEN/obj/$(basename $f).o "${HOST_SAN:-}"; fi
done
cc $HERE/emu_rt.cpp $GEN/obj/emu_rt.o -O2
for f in $GEN/src/testing/*.cpp; do cc $f $GEN/obj/testing/$(basename $f).o; done
for p in "${pids[@]}"; do wait $p || fails=1; done
[ $fails = 0 ] || { echo "tests/emu/dev/build.sh: compilation failed"; exit 1; }
$CXX -shared -fPIC $TSAN_LD -o $GEN/libzkgl.so $GEN/obj/*.o -lpthread
$CXX -shared -fPIC -o $GEN/libzkgl_testcircuits.so $GEN/obj/testing/*.o -L$GEN -lzkgl -Wl,-rpath,'$ORIGIN'
echo "built $GEN/libzkgl.so"
