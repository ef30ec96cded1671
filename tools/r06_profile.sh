#!/bin/bash
# Round-6 evidence run (one box, one call): the DRIVER's invocation three times + the W4A4 line, the same command under rocprofv3 --kernel-trace, separate --pmc FETCH_SIZE / WRITE_SIZE
# passes of the W8A8 and the W4A4 GEMM (no tracing domains beside them), in-kernel phase stamps of the metric tile, the W4A4 wait-grouping A/B.  Output: gpurun_out/r06p_* (copied to profiles/ by hand).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > $O/r06p_bench_line$i.json 2>> $O/r06p_bench.err; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --bit 4 --no-cpu-baseline > $O/r06p_bench_line_w4a4.json 2>> $O/r06p_bench.err
MIXQ_TUNING_LIB=1 timeout 300 python3 tools/prefill_sweep.py --bit 4 --tokens 512,2048 --layers 11008x4096,4096x4096 --cfgs wr128x192_s16_d4_l2,wr128x192_f6r_xw1,wr128x192_f6r_xw2,wr128x192_f6r_xw4 --rounds 15 2>&1 | grep -v amdgpu.ids > $O/r06p_f6_xwg.txt
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift; timeout 500 rocprofv3 "$@" > $O/r06p_$tag.log 2>&1; }
B="python3 $R/bench.py --gpus 1 --no-cpu-baseline --no-secondary"
prof kt --kernel-trace -d $O/prof_r06p_kt -o kt -- $B --steps 20 --warmup 5
prof kt_w4a4 --kernel-trace -d $O/prof_r06p_kt_w4a4 -o kt -- $B --steps 20 --warmup 5 --bit 4
prof fetch --pmc FETCH_SIZE -d $O/prof_r06p_fetch -o pmc -- $B --steps 20 --warmup 2 --no-graph
prof write --pmc WRITE_SIZE -d $O/prof_r06p_write -o pmc -- $B --steps 20 --warmup 2 --no-graph
prof fetch_w4a4 --pmc FETCH_SIZE -d $O/prof_r06p_fetch_w4a4 -o pmc -- $B --steps 20 --warmup 2 --no-graph --bit 4
prof write_w4a4 --pmc WRITE_SIZE -d $O/prof_r06p_write_w4a4 -o pmc -- $B --steps 20 --warmup 2 --no-graph --bit 4
cd $R
for n in kt kt_w4a4 fetch write fetch_w4a4 write_w4a4; do
    f=$(find gpurun_out/prof_r06p_$n -name "*.db" | head -1); python3 tools/rocprof_summary.py $f "" 110 > gpurun_out/r06p_$n.txt 2>&1
done
python3 tools/driver_gaps.py $(find gpurun_out/prof_r06p_kt -name "*.db" | head -1) 20 > $O/r06p_driver_gaps.txt 2>&1
rm -rf gpurun_out/prof_r06p_*
S="separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over python bench.py --steps 20 --warmup 2 --no-graph (tools/r06_profile.sh)"
python3 tools/make_traffic_json.py gpurun_out/r06p_fetch.txt gpurun_out/r06p_write.txt gpurun_out/r06p_hbm_traffic.json "$S" 512 4096 11008 8 41 > $O/r06p_traffic.log 2>&1
python3 tools/make_traffic_json.py gpurun_out/r06p_fetch_w4a4.txt gpurun_out/r06p_write_w4a4.txt gpurun_out/r06p_hbm_traffic_w4a4.json "$S --bit 4" 512 4096 11008 4 128 >> $O/r06p_traffic.log 2>&1
MIXQ_TUNING_LIB=1 python3 tools/trace_gemm.py --shapes 512x11008x4096 --cfgs wr128x192_s16_d4_l2 --nout 41 2>&1 | grep -v amdgpu.ids > $O/r06p_gemm_trace.txt
cat $O/r06p_driver_gaps.txt | tail -12
tr -s ' ' < $O/r06p_f6_xwg.txt | sed 's/ wr/\n  wr/g; s/ auto/\n  auto/g; s/ int_mm/\n  int_mm/' | grep -v "^AMD\|^us per\|auto = that\|vendor\|int_mm\|^ *$"
python3 -c "
import json
for f in ('r06p_bench_line1.json','r06p_bench_line2.json','r06p_bench_line3.json','r06p_bench_line_w4a4.json'):
    d=json.loads(open('$O/'+f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'], 'cold', d['timing'].get('cold_weights_ms_per_step'), {k:(v['tflops'], v['gemm']['us_per_launch']) for k,v in (d.get('secondary_configs') or {}).items()})
"
cat $O/r06p_traffic.log | tail -4
