#!/bin/bash
# FP6-pipe W4A4: in-kernel timeline, bench line, the two tests that failed on plumbing
export TMPDIR=/tmp
mkdir -p gpurun_out
{
MIXQ_TUNING_LIB=1 timeout 200 python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $(python - <<'PY'
import os
os.environ["MIXQ_TUNING_LIB"]="1"
from mixq_amd import _capi
print(_capi.gemm_config_names().index("wr128x192_s16_d4_l2"))
PY
) --bit 4 --f6 --nout 128
echo "== bench --bit 4"
timeout 600 python bench.py --bit 4 --no-cpu-baseline
echo "== tests"
timeout 600 python -m pytest tests/test_gpu_fp6.py tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q -k "fp6 or bench_other or operator_trace_w4 or four_bit or minus_eight or mlp_block_w4a4" 2>&1 | tail -5
} > gpurun_out/r03_f6_trace.txt 2>&1
tail -40 gpurun_out/r03_f6_trace.txt
