#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python tools/ab_libs.py --libs new=mixq_amd/libmixq_hip.so,r03prev=mixq_amd/libmixq_prev_order.so --shapes 512x11008x4096,512x4096x4096,512x8192x8192,512x28672x8192,512x6144x4096,512x4096x11008,512x14336x4096,2048x11008x4096,32x11008x4096 --nouts 41 --rounds 15 > $O/r03z_ab_order.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/r03z_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r03z_pytest.txt
grep -v amdgpu $O/r03z_ab_order.txt; tail -4 $O/r03z_pytest.txt
