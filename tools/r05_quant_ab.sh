#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q 2>&1 | tail -5 > $O/r05b_pytest.txt
cat $O/r05b_pytest.txt
{ python3 tools/time_quant_mask.py --rounds 7; python3 tools/time_quant_mask.py --rounds 7 --shape 4096,4096; python3 tools/time_quant_mask.py --rounds 7 --shape 11008,4096; python3 tools/time_quant_mask.py --rounds 7 --bit 4; } 2>&1 | grep -v amdgpu.ids > $O/r05b_quant_ab.txt
cat $O/r05b_quant_ab.txt
