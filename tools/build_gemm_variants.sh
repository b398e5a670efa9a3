#!/bin/bash
# builds build/variants/libgcn_<name>.so for tools/gemm_variant_sweep.py (runs here, on the CPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
build() { name=$1; shift; /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -I include "$@" \
  -o build/variants/libgcn_$name.so pygcn_amd/csrc/gcn_spmm.hip pygcn_amd/csrc/gcn_ingest.hip pygcn_amd/csrc/gcn_gemm.hip pygcn_amd/csrc/gcn_plan.hip & }
build base
build ring4 -DGEMM_H2_RING=4
build ring5 -DGEMM_H2_RING=5
build stage1 -DGEMM_STAGE=1
build waves4 -DGEMM_WAVES=4
wait
ls -la build/variants/
