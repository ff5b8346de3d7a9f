#!/bin/bash
# round-4 GPU collection: the whole -m gpu suite, smoke(), the bench lines that go to profiles/, rocprofv3 --stats and PMC collections (run on the GPU box from the repo root)
cd "$(dirname "$0")/.." || exit 1
ulimit -c 0
O=gpurun_out/r04z; N=gpurun_out/profiles_new; rm -rf $N; mkdir -p $O $N
R=$PWD; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
grep -E "passed|failed" $O/tests.log | tail -2 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $N/r04_bench_line_driver_form.json 2> $O/b1.err
python bench.py > $N/r04_bench_line.json 2> $O/b2.err
python bench.py --batch 1 --steps 64 --warmup 8 --no-cpu-baseline --no-extra > $N/r04_bench_line_sequential.json 2> $O/b4.err
python bench.py --steps 20 --warmup 5 --api render --no-cpu-baseline --no-extra > $N/r04_bench_line_api_render.json 2> $O/b5.err
python bench.py --config c4 --batch 32 --steps 32 --warmup 32 --no-cpu-baseline --no-extra > $N/r04_bench_line_c4_one_gpu_32_in_flight.json 2> $O/b6.err
python bench.py --config c4 --batch 16 --steps 32 --warmup 16 --no-cpu-baseline --no-extra > $N/r04_bench_line_c4_one_gpu_16_in_flight.json 2> $O/b6b.err
python bench.py --workload standin --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $N/r04_bench_line_standin_r1_r3_driver_form.json 2> $O/b7.err
python bench.py --workload standin --no-cpu-baseline --no-extra > $N/r04_bench_line_standin_r1_r3.json 2> $O/b7b.err
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/b8.err
for f in $N/r04_bench_line*.json $O/bench_n2_gloo.json; do python -c "
import json,sys
j=json.loads([l for l in open('$f') if l.startswith('{')][-1])
r=j['roofline']
print('$f'.split('/')[-1], round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), 'P', j['config']['passes_in_flight'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in j['kernel_ms_per_step'].items() if 'busy' not in k}, 'frac', round(r['frac'],3), r['bound'], j.get('value_weak'), {k: round(v['value'],1) for k,v in j.get('extra',{}).items()})
" || echo "FAILED $f"; done
# rocprofv3 --stats of the bench command lines
for cfg in "default:" "driver:--steps 20 --warmup 5"; do
  n=${cfg%%:*}; a=${cfg#*:}
  rm -rf $R/$O/stats_$n; cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/$O/stats_$n -o s -- python $R/bench.py $a --no-cpu-baseline --no-extra > $R/$O/stats_$n.log 2>&1
  cd $R
  python tools/summarize_stats.py $O/stats_$n r04_kernel_stats_$n "python bench.py $a --no-cpu-baseline --no-extra" > $O/stats_$n.txt 2>&1
  cp profiles/r04_kernel_stats_$n.md $N/
  rm -rf $O/stats_$n
done
# PMC collections, one per configuration a bench line is printed for
bash tools/collect_pmc.sh r04_pmc_bathroom2_b20 --steps 20 --warmup 5 > $O/pmc_b20.txt 2>&1
bash tools/collect_pmc.sh r04_pmc_bathroom2_b64 > $O/pmc_b64.txt 2>&1
bash tools/collect_pmc.sh r04_pmc_standin_b20 --workload standin --steps 20 --warmup 5 > $O/pmc_s20.txt 2>&1
tail -14 $O/pmc_b20.txt | cut -c1-200
# the lines again, now that the PMC summaries of their configurations exist (roofline.traffic / bound / lane_utilisation filled in)
cp $N/r04_pmc_*.json profiles/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $N/r04_bench_line_driver_form.json 2> $O/b1.err
python bench.py > $N/r04_bench_line.json 2> $O/b2.err
# the widened rows
python bench.py --renderer bpt --no-cpu-baseline > $N/r04_bench_line_bpt.json 2> $O/w1.err
python bench.py --renderer psfpt --no-cpu-baseline > $N/r04_bench_line_psfpt.json 2> $O/w2.err
for f in driver_form "" _bpt _psfpt; do python -c "
import json
g='$N/r04_bench_line$f.json' if '$f' in ('', '_bpt', '_psfpt') else '$N/r04_bench_line_$f.json'
j=json.loads([l for l in open(g) if l.startswith('{')][-1]); r=j['roofline']
print(g.split('/')[-1], round(j['value'],1), j['config']['passes_in_flight'], j['kernel_ms_per_step'], 'frac', round(r['frac'],3), r['bound'], r.get('lane_utilisation'), r.get('counter_frac'))"; done
python tools/emulate_shares.py > $O/shares.txt 2>&1; tail -8 $O/shares.txt | cut -c1-220
python tools/diag_launches.py --workload bathroom2 --batch 1 4 20 64 > $N/r04_launch_times_bathroom2.txt 2>&1
python tools/diag_launches.py --workload standin --batch 1 20 64 > $N/r04_launch_times_standin_r1_r3.txt 2>&1
rm -rf $R/gpurun_out/pmc
ls $N
