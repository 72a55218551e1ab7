#!/bin/bash
# tests/emu/build.sh [variant [defs]] — build the LANE HARNESS (tests/emu/README.md) against libzkgl.so (or against a side-by-side library libzkgl_<variant>.so
# built with the SAME compile-time switches, e.g. an elimination-probe build)
# -> tests/emu/_gen/libzkgl_emu[_<variant>].so     (host clang of the ROCm toolchain: the device headers use ext_vector_type)
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
V=${1:-}; shift || true
GEN=$HERE/_gen; mkdir -p $GEN
LIB=zkgl; OUT=$GEN/libzkgl_emu.so
if [ -n "$V" ]; then LIB=zkgl_$V; OUT=$GEN/libzkgl_emu_$V.so; fi
# up to date? (the product library it links against, the device headers it is cut from, its own sources)
NEWEST=$(ls -t $ROOT/era-zkevm_circuits_amd/lib$LIB.so $ROOT/era-zkevm_circuits_amd/csrc/*.hpp $HERE/*.cpp $HERE/*.hpp $HERE/*.py $HERE/build.sh | head -1)
if [ -f $OUT ] && [ $OUT -nt $NEWEST ] && [ "$(cat $OUT.flags 2>/dev/null)" = "$*" ]; then echo "up to date $OUT"; exit 0; fi
python $HERE/gen.py $GEN
/opt/rocm/lib/llvm/bin/clang++ -std=c++20 -O1 -fPIC -shared -Wno-unknown-attributes -Wno-ignored-attributes -Wno-macro-redefined -Wno-unused-value -Wno-pass-failed -Wno-keyword-macro \
  -I$HERE/stub -I$GEN -I$HERE "$@" $HERE/emu_harness.cpp -o $OUT -L$ROOT/era-zkevm_circuits_amd -l$LIB -Wl,-rpath,$ROOT/era-zkevm_circuits_amd -lpthread
echo "$*" > $OUT.flags
echo "built $OUT"
