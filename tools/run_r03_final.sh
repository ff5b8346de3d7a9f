#!/bin/bash
# round-3 final GPU call: the whole -m gpu suite, smoke(), the bench lines that go to profiles/, rocprofv3 --stats and PMC collections
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03z; N=gpurun_out/profiles_new; mkdir -p $O $N
R=$PWD; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
tail -4 $O/tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $N/r03_bench_line_driver_form.json 2> $O/b1.err
python bench.py > $N/r03_bench_line.json 2> $O/b2.err
python bench.py --workload testball-room --no-cpu-baseline > $N/r03_bench_line_testball_room.json 2> $O/b3.err
python bench.py --batch 1 --steps 64 --warmup 8 --no-cpu-baseline --no-extra > $N/r03_bench_line_sequential.json 2> $O/b4.err
python bench.py --steps 20 --warmup 5 --api render --no-cpu-baseline --no-extra > $N/r03_bench_line_api_render.json 2> $O/b5.err
python bench.py --config c4 --steps 16 --warmup 16 --no-cpu-baseline > $N/r03_bench_line_c4_one_gpu.json 2> $O/b6.err
python bench.py --steps 20 --warmup 5 --lanes 2 --no-cpu-baseline --no-extra > $N/r03_bench_line_driver_form_lanes2.json 2> $O/b7.err
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/b8.err
for f in $N/r03_bench_line*.json $O/bench_n2_gloo.json; do python -c "
import json,sys
j=json.loads([l for l in open('$f') if l.startswith('{')][-1])
r=j['roofline']
print('$f'.split('/')[-1], round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), 'P', j['config']['passes_in_flight'], 'lanes', j['config']['render_lanes'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in j['kernel_ms_per_step'].items()}, 'frac', round(r['frac'],3), j.get('value_weak'), j.get('speedup_vs_n1'))
" || echo "FAILED $f"; done
# rocprofv3 --stats of the bench command lines
for cfg in "default:" "driver:--steps 20 --warmup 5"; do
  n=${cfg%%:*}; a=${cfg#*:}
  rm -rf $R/$O/stats_$n; cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/$O/stats_$n -o s -- python $R/bench.py $a --no-cpu-baseline --no-extra > $R/$O/stats_$n.log 2>&1
  cd $R
  python tools/summarize_stats.py $O/stats_$n r03_kernel_stats_$n "python bench.py $a --no-cpu-baseline --no-extra" > $O/stats_$n.txt 2>&1
  cp profiles/r03_kernel_stats_$n.md $N/
  rm -rf $O/stats_$n
done
# PMC collections, one per configuration a bench line is printed for
bash tools/collect_pmc.sh r03_pmc_standin_b20 --steps 20 --warmup 5 > $O/pmc_b20.txt 2>&1
bash tools/collect_pmc.sh r03_pmc_standin_b64 > $O/pmc_b64.txt 2>&1
bash tools/collect_pmc.sh r03_pmc_testball_b64 --workload testball-room > $O/pmc_tb.txt 2>&1
tail -12 $O/pmc_b20.txt | cut -c1-200
# the widened rows
python bench.py --renderer bpt --sc 0 --no-cpu-baseline > $N/r03_bench_line_bpt_sc0.json 2> $O/w3.err
bash tools/collect_pmc.sh r03_pmc_bpt_sc1_b32 --renderer bpt > $O/pmc_bpt.txt 2>&1
bash tools/collect_pmc.sh r03_pmc_psfpt_b32 --renderer psfpt > $O/pmc_psfpt.txt 2>&1
python bench.py --renderer bpt --no-cpu-baseline > $N/r03_bench_line_bpt.json 2> $O/w1.err
python bench.py --renderer psfpt --no-cpu-baseline > $N/r03_bench_line_psfpt.json 2> $O/w2.err
for f in bpt bpt_sc0 psfpt; do python -c "
import json
j=json.loads([l for l in open('$N/r03_bench_line_$f.json') if l.startswith('{')][-1]); r=j['roofline']
print('$f', round(j['value'],1), j['config']['passes_in_flight'], j['kernel_ms_per_step'], 'frac', round(r['frac'],3))"; done
# the EAW filter: PMC evidence for what bounds it
D=$R/$O/pmc_filter; rm -rf $D; mkdir -p $D; cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o p -- python $R/tools/bench_filter.py > $R/$O/filter_line.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o p -- python $R/tools/bench_filter.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $D/valu -o p -- python $R/tools/bench_filter.py > /dev/null 2>&1
cd $R
python tools/summarize_pmc_kernels.py $D r03_pmc_filter "python tools/bench_filter.py (fpt_filter at 1600x900: 14 a-trous steps + 2 variance filters)" "fpt::" > $O/pmc_filter.txt 2>&1
cp profiles/r03_pmc_filter.json $N/ 2>/dev/null
tail -3 $O/filter_line.txt | cut -c1-300
rm -rf $D $R/gpurun_out/pmc
ls $N
