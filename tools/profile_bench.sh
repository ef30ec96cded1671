R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01c_kt -o kt -- python $R/bench.py --steps 200 --warmup 20 > $R/gpurun_out/prof_r01c_kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r01c_fetch -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph > $R/gpurun_out/prof_r01c_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r01c_write -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph > $R/gpurun_out/prof_r01c_write.log 2>&1
cd $R
for n in kt fetch write; do f=$(find gpurun_out/prof_r01c_$n -name "*.db" | head -1); python tools/rocprof_summary.py $f > gpurun_out/prof_r01c_$n.txt 2>&1; done
tail -1 gpurun_out/prof_r01c_kt.log | cut -c1-300
head -6 gpurun_out/prof_r01c_kt.txt
grep -A3 "gemm_kernel" gpurun_out/prof_r01c_fetch.txt | tail -5
grep -A3 "gemm_kernel" gpurun_out/prof_r01c_write.txt | tail -5
python bench.py > gpurun_out/bench_r01c.json 2> gpurun_out/bench_r01c.err; tail -c 600 gpurun_out/bench_r01c.json
