#!/bin/bash
# Round-4 evidence run on the GPU box (one box, one call): everything lands in gpurun_out/r04p_*; copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() { # tag, rocprof args..., -- bench args
    tag=$1; shift
    timeout 400 rocprofv3 "$@" > $O/r04p_$tag.log 2>&1
}
B="python $R/bench.py --no-cpu-baseline --no-secondary"
prof kt --kernel-trace --stats -d $O/prof_r04p_kt -o kt -- $B --steps 200 --warmup 20
for cfg in "" "--bit 4" "--shape 8192,28672"; do
    t=$(echo "$cfg" | tr -d ' ,-' ); t=${t:-metric}
    prof fetch_$t --pmc FETCH_SIZE -d $O/prof_r04p_fetch_$t -o pmc -- $B --steps 20 --warmup 2 --no-graph $cfg
    prof write_$t --pmc WRITE_SIZE -d $O/prof_r04p_write_$t -o pmc -- $B --steps 20 --warmup 2 --no-graph $cfg
done
prof mfma --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE -d $O/prof_r04p_mfma -o pmc -- $B --steps 20 --warmup 2 --no-graph
prof mem --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/prof_r04p_mem -o pmc -- $B --steps 20 --warmup 2 --no-graph
prof kt_block --kernel-trace --stats -d $O/prof_r04p_kt_block -o kt -- python $R/tools/bench_block.py --layers 32 --passes 1
cd $R
for n in kt fetch_metric write_metric fetch_bit4 write_bit4 fetch_shape819228672 write_shape819228672 mfma mem kt_block; do
    f=$(find gpurun_out/prof_r04p_$n -name "*.db" | head -1); python tools/rocprof_summary.py $f > gpurun_out/r04p_$n.txt 2>&1
done
rm -rf gpurun_out/prof_r04p_*
NOTE="separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over python bench.py --steps 20 --warmup 2 --no-graph (tools/r04_profile.sh)"
python tools/make_traffic_json.py gpurun_out/r04p_fetch_metric.txt gpurun_out/r04p_write_metric.txt gpurun_out/r04p_hbm_traffic.json "$NOTE" 512 4096 11008 8 41 > gpurun_out/r04p_traffic.log 2>&1
python tools/make_traffic_json.py gpurun_out/r04p_fetch_bit4.txt gpurun_out/r04p_write_bit4.txt gpurun_out/r04p_hbm_traffic_w4a4.json "$NOTE --bit 4" 512 4096 11008 4 128 >> gpurun_out/r04p_traffic.log 2>&1
python tools/make_traffic_json.py gpurun_out/r04p_fetch_shape819228672.txt gpurun_out/r04p_write_shape819228672.txt gpurun_out/r04p_hbm_traffic_8192x28672.json "$NOTE --shape 8192,28672" 512 8192 28672 8 82 >> gpurun_out/r04p_traffic.log 2>&1
mkdir -p profiles_tmp && cp gpurun_out/r04p_hbm_traffic*.json profiles/ 2>/dev/null; for f in profiles/r04p_hbm_traffic*.json; do mv $f ${f/r04p_/r04_}; done; rmdir profiles_tmp
python bench.py > gpurun_out/r04p_bench.json 2> gpurun_out/r04p_bench.err
python bench.py --bit 4 --no-cpu-baseline > gpurun_out/r04p_bench_w4.json 2>> gpurun_out/r04p_bench.err
python bench.py --shape 8192,28672 --no-cpu-baseline --steps 100 > gpurun_out/r04p_bench_70b.json 2>> gpurun_out/r04p_bench.err
cp profiles/r04_hbm_traffic*.json gpurun_out/ 2>/dev/null
# the secondary tables (no profiler attached), same box
python tools/bench_mlp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_mlp_block.txt
python tools/bench_block.py --layers 32 --passes 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_block.txt
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_configs_operator.txt
{ for t in 16 32 64 128 256 512 1024 2048 4096; do python tools/bench_pair.py --tokens $t --rounds 5; done
  python tools/bench_pair.py --tokens 512 --amax --rounds 5
  python tools/bench_pair.py --tokens 512 --shape 8192,28672 --rounds 5
  python tools/bench_pair.py --tokens 512 --shape 4096,14336 --rounds 5
  for t in 16 128 512 2048; do python tools/bench_pair.py --bit 4 --outliers 128 --tokens $t --rounds 5; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_pair_launch_ab.txt
{ python tools/time_quant_mask.py; python tools/time_quant_mask.py --bit 4; python tools/time_quant_mask.py --shape 4096,4096; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_quant_kept_mask.txt
python tools/yardstick.py --no-power --rounds 10 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_ceiling.txt
head -8 gpurun_out/r04p_kt.txt; tail -12 gpurun_out/r04p_traffic.log; tail -c 1500 gpurun_out/r04p_bench.json; echo; tail -c 700 gpurun_out/r04p_bench_w4.json; echo; tail -c 700 gpurun_out/r04p_bench_70b.json
