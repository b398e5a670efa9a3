#!/bin/bash
# builds build/variants/libgcn_<name>.so — cache-policy experiments of the SpMM gather
# (tools/spmm_policy_sweep.py; runs here, on the CPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants build/variants/obj
rm -f build/variants/*.so
HIPCC="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I include"
for f in gcn_ingest gcn_gemm gcn_plan gcn_pack; do
  [ -f pygcn_amd/csrc/build/$f.hip.o ] || { echo "run python -m pygcn_amd.build first"; exit 1; }
done
build() { name=$1; shift; ( $HIPCC "$@" -c pygcn_amd/csrc/gcn_spmm.hip -o build/variants/obj/spmm_$name.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libgcn_$name.so build/variants/obj/spmm_$name.o \
  pygcn_amd/csrc/build/gcn_ingest.hip.o pygcn_amd/csrc/build/gcn_gemm.hip.o pygcn_amd/csrc/build/gcn_plan.hip.o pygcn_amd/csrc/build/gcn_pack.hip.o ) & }
build base
build allnt -DSPMM_GATHER_AUX=2
build storent -DSPMM_STORE_NT=1
build hub -DSPMM_HUB_TAG=1 -DSPMM_GATHER_AUX=2
build hubstore -DSPMM_HUB_TAG=1 -DSPMM_GATHER_AUX=2 -DSPMM_STORE_NT=1
wait
ls -la build/variants/
