#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
NAMES=$(python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('128x192_w2x2_s5_l4','wr128x192_s8_d4_l1','wr128x192_s8_d4_l2','wr128x192_s8_d6_l1','wr128x192_abl1_noW','wr128x192_abl2_noX','wr128x192_abl3_mfma')))")
timeout 300 python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $NAMES > $O/r02c_trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass -d $R/$O/prof_r02c_$tag -o pmc -- python $R/tools/sweep_gemm.py --shapes 512x11008x4096 --cfgs $NAMES --packed 1,2 --iters 10 --out $R/$O/tmp.json > $R/$O/r02c_pmc_$tag.log 2>&1
  f=$(find $R/$O/prof_r02c_$tag -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $f gemm > $R/$O/r02c_pmc_$tag.txt 2>&1
done
cd $R; cat $O/r02c_trace.txt | head -120
