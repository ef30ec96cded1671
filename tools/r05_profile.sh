#!/bin/bash
# Round-5 evidence run (one box, one call): the DRIVER's invocation un-profiled and under rocprofv3 --kernel-trace (gaps per step), the
# memory-bound family's kernel durations inside the real forwards + PMC passes (separate --pmc runs, no tracing domains beside them),
# HBM traffic of the GEMM.  Everything lands in gpurun_out/r05p_*; copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > $O/r05p_bench_line$i.json 2>> $O/r05p_bench.err; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --bit 4 --no-cpu-baseline > $O/r05p_bench_line_w4a4.json 2>> $O/r05p_bench.err
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift; timeout 500 rocprofv3 "$@" > $O/r05p_$tag.log 2>&1; }
B="python3 $R/bench.py --gpus 1 --no-cpu-baseline --no-secondary"
prof kt --kernel-trace -d $O/prof_r05p_kt -o kt -- $B --steps 20 --warmup 5
prof kt_k11008 --kernel-trace -d $O/prof_r05p_kt_k11008 -o kt -- $B --steps 20 --warmup 5 --shape 11008,4096
prof kt_mlp --kernel-trace -d $O/prof_r05p_kt_mlp -o kt -- python3 $R/tools/bench_mlp.py
Q="python3 $R/tools/quant_family.py --eager 12"
prof q_fetch --pmc FETCH_SIZE -d $O/prof_r05p_q_fetch -o pmc -- $Q
prof q_write --pmc WRITE_SIZE -d $O/prof_r05p_q_write -o pmc -- $Q
prof q_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $O/prof_r05p_q_sq -o pmc -- $Q
prof q_inst --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM -d $O/prof_r05p_q_inst -o pmc -- $Q
prof fetch --pmc FETCH_SIZE -d $O/prof_r05p_fetch -o pmc -- $B --steps 20 --warmup 2 --no-graph
prof write --pmc WRITE_SIZE -d $O/prof_r05p_write -o pmc -- $B --steps 20 --warmup 2 --no-graph
cd $R
for n in kt kt_k11008 kt_mlp q_fetch q_write q_sq q_inst fetch write; do
    f=$(find gpurun_out/prof_r05p_$n -name "*.db" | head -1); python3 tools/rocprof_summary.py $f "" 110 > gpurun_out/r05p_$n.txt 2>&1
done
python3 tools/driver_gaps.py $(find gpurun_out/prof_r05p_kt -name "*.db" | head -1) 20 > $O/r05p_driver_gaps.txt 2>&1
rm -rf gpurun_out/prof_r05p_*
python3 tools/make_traffic_json.py gpurun_out/r05p_fetch.txt gpurun_out/r05p_write.txt gpurun_out/r05p_hbm_traffic.json "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over python bench.py --steps 20 --warmup 2 --no-graph (tools/r05_profile.sh)" 512 4096 11008 8 41 > $O/r05p_traffic.log 2>&1
python3 tools/trace_gemm.py --shapes 512x11008x4096 --cfgs wr128x192_s16_d4_l2 --nout 41 2>&1 | grep -v amdgpu.ids > $O/r05p_gemm_trace.txt
python3 tools/trace_gemm.py --shapes 512x4096x4096,512x4096x11008 --cfgs wr64x128_s16_d4_l2 --nout 41 2>&1 | grep -v amdgpu.ids >> $O/r05p_gemm_trace.txt
cat $O/r05p_driver_gaps.txt
grep -h "quant\|rmsnorm" $O/r05p_kt.txt $O/r05p_kt_k11008.txt $O/r05p_kt_mlp.txt | head -20
python3 -c "
import json
for f in ('r05p_bench_line1.json','r05p_bench_line2.json','r05p_bench_line3.json','r05p_bench_line_w4a4.json'):
    d=json.loads(open('$O/'+f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'])
"
