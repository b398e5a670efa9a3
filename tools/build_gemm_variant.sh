#!/bin/bash
# One experiment build of the library with extra hipcc flags for gcn_gemm.hip only (the other objects
# are the product build's): tools/build_gemm_variant.sh <name> <flags...>  ->  build/variants/libgcn_<name>.so
# (runs here, on the CPU box; the .so travels to the GPU box with the snapshot; GCN_SPMM_LIB selects it)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
python -m pygcn_amd.build >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-inline-asm -I include "$@" \
  -c pygcn_amd/csrc/gcn_gemm.hip -o build/variants/gcn_gemm_$name.o
objs=$(ls pygcn_amd/csrc/build/*.o | grep -v gcn_gemm)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libgcn_$name.so build/variants/gcn_gemm_$name.o $objs
ls -la build/variants/libgcn_$name.so
