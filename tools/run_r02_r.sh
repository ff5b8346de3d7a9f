#!/bin/bash
# the widened rows' profiles: PMC collections (config-matched traffic for their bench lines), bench lines, rocprofv3 --stats
mkdir -p gpurun_out/r02r gpurun_out/profiles_new
bash tools/collect_pmc.sh r02_pmc_bpt_sc1_b32 --renderer bpt > gpurun_out/r02r/pmc_bpt.txt 2>&1
bash tools/collect_pmc.sh r02_pmc_psfpt_b32 --renderer psfpt > gpurun_out/r02r/pmc_psfpt.txt 2>&1
python bench.py --renderer bpt --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_bpt.json 2> gpurun_out/r02r/b1.err
python bench.py --renderer psfpt --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_psfpt.json 2> gpurun_out/r02r/b2.err
python bench.py --renderer psfpt --batch 1 --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_psfpt_sequential.json 2> gpurun_out/r02r/b2b.err
python bench.py --renderer bpt --sc 0 --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_bpt_sc0.json 2> gpurun_out/r02r/b3.err
for f in bpt psfpt psfpt_sequential bpt_sc0; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/profiles_new/r02_bench_line_$f.json') if l.startswith('{')][-1]); r=j['roofline']
print('$f', round(j['value'],1), j['config']['passes_in_flight'], j['kernel_ms_per_step'], 'frac', round(r['frac'],3), 'traffic', r['traffic'], 'counter_frac', r['counter_frac'], r['traffic_source'][:60])"; done
R=$PWD; export TMPDIR=/tmp
for cfg in "bpt:--renderer bpt" "psfpt:--renderer psfpt"; do
  n=${cfg%%:*}; a=${cfg#*:}
  rm -rf $R/gpurun_out/r02r/stats_$n; cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02r/stats_$n -o s -- python $R/bench.py $a --no-cpu-baseline > $R/gpurun_out/r02r/stats_$n.log 2>&1
  cd $R
  python tools/summarize_stats.py gpurun_out/r02r/stats_$n r02_kernel_stats_$n "python bench.py $a --no-cpu-baseline" > gpurun_out/r02r/stats_$n.txt 2>&1
  cp profiles/r02_kernel_stats_$n.md gpurun_out/profiles_new/
  rm -rf gpurun_out/r02r/stats_$n
  head -22 gpurun_out/r02r/stats_$n.txt | cut -c1-160
done
