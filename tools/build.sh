#!/bin/bash
# Build libmixq_hip.so for gfx950 (same command __graft_entry__.build() runs).
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o mixq_amd/libmixq_hip.so.tmp mixq_amd/csrc/quant.hip mixq_amd/csrc/gemm.hip mixq_amd/csrc/gemm_sk.hip mixq_amd/csrc/norm.hip mixq_amd/csrc/gemm_w8a16.hip mixq_amd/csrc/gemm_skinny.hip "$@"
mv mixq_amd/libmixq_hip.so.tmp mixq_amd/libmixq_hip.so
echo "built mixq_amd/libmixq_hip.so"
