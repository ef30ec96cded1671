#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
{
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p70_touch --launches 24
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p70_touch --launches 24 --cold 8
MIXQ_TUNING_LIB=1 python - <<'PY'
import torch
from mixq_amd import _capi, mixlib
lib=_capi.load(); names=_capi.gemm_config_names()
g=torch.Generator().manual_seed(1)
M,N,K=512,11008,4096
qx=torch.randint(-127,128,(M,K),generator=g,dtype=torch.int8).cuda(); qw=torch.randint(-127,128,(N,K),generator=g,dtype=torch.int8).cuda()
sx=(torch.rand(M,1,generator=g)*0.01+0.001).half().cuda(); sw=(torch.rand(1,N,generator=g)*0.01+0.001).half().cuda()
xp=mixlib.PackOperand(qx,1); wp=mixlib.PackOperand(qw,2)
outs=[]
for nm in ('wr128x192_s16_d4_l2','wr128x192_p70_touch'):
    lib.mixq_gemm_set_config(names.index(nm)); y=mixlib.FusedLinear(xp,wp,sx,sw,None,None,0,None,M,N,K,bit=8); torch.cuda.synchronize(); outs.append(y.clone())
lib.mixq_gemm_set_config(-1)
print('touch variant bit-identical:', bool(torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))))
PY
} 2>&1 | grep -v amdgpu > $O/r04v_touch.txt
cat $O/r04v_touch.txt
