#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
MIXQ_TUNING_LIB=1 python - > $O/r04j_direct_check.txt 2>&1 <<'PY'
# direct-store epilogue (probe): bit-identical to the staged one?
import torch, numpy as np
from mixq_amd import _capi, mixlib
lib=_capi.load(); names=_capi.gemm_config_names()
torch.manual_seed(0)
for (M,N,K,nout) in [(512,11008,4096,41),(130,200,512,16),(512,384,256,0),(257,1000,1024,41)]:
    g=torch.Generator().manual_seed(1)
    qx=torch.randint(-127,128,(M,K),generator=g,dtype=torch.int8).cuda(); qw=torch.randint(-127,128,(N,K),generator=g,dtype=torch.int8).cuda()
    sx=(torch.rand(M,1,generator=g)*0.01+0.001).half().cuda(); sw=(torch.rand(1,N,generator=g)*0.01+0.001).half().cuda()
    xp=mixlib.PackOperand(qx,1); wp=mixlib.PackOperand(qw,2)
    xo=wo=None
    if nout:
        pad=(nout+15)//16*16
        xo=torch.randn((M,pad),device='cuda').half()[:,:nout]; wo=torch.randn((N,pad),device='cuda').half()[:,:nout]
    outs={}
    for nm in ('wr128x192_s16_d4_l2','wr128x192_p63_epi2_direct','wr128x192_p60_epi1'):
        lib.mixq_gemm_set_config(names.index(nm))
        y=mixlib.FusedLinear(xp,wp,sx,sw,xo,wo,nout,None,M,N,K,bit=8)
        torch.cuda.synchronize(); outs[nm]=y.clone()
    lib.mixq_gemm_set_config(-1)
    a=outs['wr128x192_s16_d4_l2']
    print((M,N,K,nout), {k: bool(torch.equal(a.view(torch.int16),v.view(torch.int16))) for k,v in outs.items()})
PY
cat $O/r04j_direct_check.txt
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1,wr128x192_p63_epi2_direct,wr128x192_abl3_mfma > $O/r04j_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_p63_epi2_direct',)))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 > $O/r04j_trace.txt 2>&1
cat $O/r04j_ab.txt $O/r04j_trace.txt
