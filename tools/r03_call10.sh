#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -m gpu -q -k "norm or mlp or fused" > $O/r03k_pytest_norm.txt 2>&1; echo "rc=$?" >> $O/r03k_pytest_norm.txt
rm -f $O/r03k_norm_ab.txt
for rep in 1 2 3; do
  echo "== round-2 kernel (256 threads per row) rep $rep" >> $O/r03k_norm_ab.txt; MIXQ_LIB_FILE=libmixq_prev_norm256.so python tools/time_norm.py 2>&1 | grep -v amdgpu >> $O/r03k_norm_ab.txt
  echo "== round-3 kernel (512 threads per row, one round trip) rep $rep" >> $O/r03k_norm_ab.txt; python tools/time_norm.py 2>&1 | grep -v amdgpu >> $O/r03k_norm_ab.txt
done
python tools/bench_mlp.py 2>&1 | grep -v amdgpu > $O/r03k_mlp.txt
tail -4 $O/r03k_pytest_norm.txt; grep "M=512 K=4096" $O/r03k_norm_ab.txt; cat $O/r03k_mlp.txt
