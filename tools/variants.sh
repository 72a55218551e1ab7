#!/bin/bash
# container: build side-by-side kernel variants of libzkgl (only zkgl_device.hip differs), in parallel.
# usage: tools/variants.sh TAG="-DFLAG ..." [TAG2="..."] ...   -> era-zkevm_circuits_amd/libzkgl_var_<TAG>.so
cd "$(dirname "$0")/../era-zkevm_circuits_amd/csrc"
mkdir -p ../build/var
pids=()
for kv in "$@"; do
  tag=${kv%%=*}; defs=${kv#*=}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $defs -c zkgl_device.hip -o ../build/var/dev_$tag.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libzkgl_var_$tag.so ../build/var/dev_$tag.o $(ls ../build/*.o | grep -v zkgl_device) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib && echo "built $tag" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
