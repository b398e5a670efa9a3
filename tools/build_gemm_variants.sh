#!/bin/bash
# builds build/variants/libgcn_<name>.so for tools/gemm_variant_sweep.py (runs here, on the CPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
rm -f build/variants/*.so
build() { name=$1; shift; /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -I include "$@" \
  -o build/variants/libgcn_$name.so pygcn_amd/csrc/gcn_spmm.hip pygcn_amd/csrc/gcn_ingest.hip pygcn_amd/csrc/gcn_gemm.hip pygcn_amd/csrc/gcn_plan.hip & }
if [ "$1" = "atg" ]; then      # ablation builds of gcn_gemm_atg256_f32 for tools/atg_variant_sweep.py
  build base
  build noload -DATG_ABLATE=4      # the step loop without its loads: what the arithmetic alone costs
else
  build base -DGEMM_H2_XLDS=0
  build dma -DGEMM_H2_XLDS=1
fi
wait
ls -la build/variants/
