#!/bin/bash
# Host-side AddressSanitizer pass over the C ABI (tools/asan_capi.cpp).  Builds here or on the GPU box; runs on the GPU box.
#   tools/asan_capi.sh build     compile the ASAN library + driver (no GPU needed)
#   tools/asan_capi.sh run       run the driver (GPU), output -> gpurun_out/r05_asan_capi.txt
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
H=/opt/rocm/bin/hipcc
SAN="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1"
if [ "$1" = build ]; then
  make -C mixq_amd/csrc -j8 LIB=../libmixq_hip_asan.so OBJDIR=../../build/obj_asan FLAGS="--offload-arch=gfx950 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 $SAN" > build/asan_build.log 2>&1 || { tail -20 build/asan_build.log; exit 1; }
  $H --offload-arch=gfx950 -std=c++17 $SAN -o build/asan_capi tools/asan_capi.cpp -Lmixq_amd -l:libmixq_hip_asan.so -Wl,-rpath,'$ORIGIN/../mixq_amd'
  echo "built build/asan_capi + mixq_amd/libmixq_hip_asan.so"
else
  mkdir -p gpurun_out
  # (the ROCm runtime maps memory inside ASAN's shadow gap: protect_shadow_gap=0; its process-lifetime allocations are not leaks of ours)
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:halt_on_error=0 ./build/asan_capi > gpurun_out/r05_asan_capi.txt 2>&1; rc=$?
  echo "exit code $rc" >> gpurun_out/r05_asan_capi.txt
  tail -15 gpurun_out/r05_asan_capi.txt
fi
