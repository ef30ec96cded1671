#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
MIXQ_TUNING_LIB=1 timeout 600 python tools/yardstick.py --energy --shapes 512x11008x4096,4096x11008x4096 --power-seconds 3 > $O/r04r_energy.txt 2>&1
grep -v amdgpu.ids $O/r04r_energy.txt
timeout 900 python tools/yardstick.py --shapes 512x11008x4096,2048x11008x4096,4096x11008x4096,8192x11008x4096,512x28672x8192,4096x4096x4096 --no-power > $O/r04r_ceiling.txt 2>&1
grep -v amdgpu.ids $O/r04r_ceiling.txt
