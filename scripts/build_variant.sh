#!/bin/bash
# Build an experimental variant of libmi355x_hotpath.so: one source recompiled with extra -D flags.
# usage: scripts/build_variant.sh NAME SOURCE.hip -DFLAG...   ->  variants/libNAME.so
# run with MI355X_HOTPATH_LIB=variants/libNAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p variants
obj=variants/${name}_$(basename "$src" .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -Iinclude -Wno-inline-asm "$@" -c "vllm_metax_amd/csrc/$src" -o "$obj"
others=$(ls vllm_metax_amd/csrc/_obj/*.o | grep -v "/$(basename "$src" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "variants/lib${name}.so" "$obj" $others
echo "built variants/lib${name}.so"
