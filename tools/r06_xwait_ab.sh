#!/bin/bash
# Round 6: hand-counted activation fragment reads (asm ds_read + lgkmcnt(n)) against the previous build's loop (compiler-placed lgkmcnt(0)),
# two libraries interleaved on one box; then the GPU suite on the new library and the scalar-cache prefetch micro-benchmark.
# mixq_amd/libmixq_hip.so.r5loop (git-ignored) = the product objects with gemm_wreg.hip of commit 1f38e8a:  git show 1f38e8a:mixq_amd/csrc/gemm_wreg.hip > /tmp/old/gemm_wreg.hip;
#   hipcc <Makefile FLAGS> -Imixq_amd/csrc -Iinclude -c /tmp/old/gemm_wreg.hip -o /tmp/old/gemm_wreg_old.o;  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o mixq_amd/libmixq_hip.so.r5loop build/obj/{quant,gemm,norm,gemm_w8a16,gemm_skinny,forward}.o /tmp/old/gemm_wreg_old.o
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
L="new=mixq_amd/libmixq_hip.so,r5loop=mixq_amd/libmixq_hip.so.r5loop"
timeout 300 python3 tools/ab_libs.py --libs $L --shapes 512x11008x4096,512x4096x4096,512x4096x11008,512x14336x4096,512x28672x8192,2048x11008x4096,4096x11008x4096 --nouts 41 --rounds 15 2>&1 | grep -v amdgpu.ids > $O/r06f_ab_i8.txt; cat $O/r06f_ab_i8.txt
timeout 300 python3 tools/ab_libs.py --bit 4 --libs $L --shapes 512x11008x4096,512x4096x4096,512x4096x11008,2048x11008x4096 --nouts 128 --rounds 15 2>&1 | grep -v amdgpu.ids > $O/r06f_ab_i4.txt; cat $O/r06f_ab_i4.txt
timeout 900 python3 -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06f_suite.txt 2>&1; echo "suite rc=$?"; grep -v amdgpu.ids $O/r06f_suite.txt | tail -8
timeout 200 tools/ubench_sprefetch > $O/r06f_sprefetch.txt 2>&1; cat $O/r06f_sprefetch.txt
