#!/bin/bash
# Builds libzkgl.so (gfx950 only) in-tree.  Usage: era-zkevm_circuits_amd/build.sh
set -euo pipefail
cd "$(dirname "$0")/csrc"
OUT=${ZKGL_OUT:-../libzkgl.so}   # ZKGL_OUT / ZKGL_DEFS / ZKGL_BUILD_DIR: side-by-side variants (host + device) for A/B runs
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value ${ZKGL_DEFS:-}"
BUILD=${ZKGL_BUILD_DIR:-../build}
mkdir -p $BUILD
pids=()
OBJS=()   # the library is linked from THIS list (not $BUILD/*.o: an incremental build dir may hold objects of sources that have moved away)
# incremental: an object is rebuilt when its source, any header of the tree or the flags changed (ZKGL_REBUILD=1 forces everything)
STAMP=$BUILD/.flags
if [ "$(cat $STAMP 2>/dev/null)" != "$FLAGS" ]; then rm -f $BUILD/*.o; echo "$FLAGS" > $STAMP; fi
NEWEST_HDR=$(ls -t *.hpp circuits/*.hpp ../../include/*.h | head -1)
stale() {  # $1 source, $2 object
  [ -n "${ZKGL_REBUILD:-}" ] || [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ "$NEWEST_HDR" -nt "$2" ]
}
for f in zkgl_device.hip; do
  o=$BUILD/$(basename $f).o; OBJS+=($o)
  if stale $f $o; then hipcc $FLAGS -c $f -o $o & pids+=($!); fi
done
for f in comm.cpp host_pool.cpp witness_pack.cpp vm_pack.cpp cs.cpp cs_perm.cpp ntt.cpp gadgets.cpp poseidon_consts.cpp capi.cpp circuits/ram_permutation.cpp circuits/main_vm.cpp circuits/opcode_defs.cpp circuits/storage_validity.cpp circuits/log_sorter.cpp circuits/keccak.cpp circuits/sha256.cpp circuits/eip4844.cpp circuits/demux_log_queue.cpp circuits/sort_decommits.cpp circuits/code_unpacker.cpp circuits/linear_hasher.cpp; do
  o=$BUILD/$(basename $f).o; OBJS+=($o)
  if stale $f $o; then hipcc $FLAGS -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $f -o $o & pids+=($!); fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT "${OBJS[@]}" -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $(readlink -f $OUT)"
# test-only circuits: their own library beside the product's (links against it; tests load it through zkgl.testlib())
if [ "$(basename $OUT)" = "libzkgl.so" ]; then
  TB=$BUILD/testing; mkdir -p $TB
  tp=()
  for f in testing/vm_shaped.cpp testing/test_capi.cpp; do
    o=$TB/$(basename $f).o
    if stale $f $o; then hipcc $FLAGS -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $f -o $o & tp+=($!); fi
  done
  for p in "${tp[@]}"; do wait $p; done
  hipcc -shared -fPIC -o $(dirname $OUT)/libzkgl_testcircuits.so $TB/*.o -L$(dirname $OUT) -lzkgl -Wl,-rpath,'$ORIGIN'
fi
