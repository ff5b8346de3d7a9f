#!/bin/bash
# PMC collections of the widened rows (BPT -sc 1, PSFPT): config-matched traffic for their bench lines
mkdir -p gpurun_out/r02r gpurun_out/profiles_new
bash tools/collect_pmc.sh r02_pmc_bpt_sc1_b16 --renderer bpt > gpurun_out/r02r/pmc_bpt.txt 2>&1
bash tools/collect_pmc.sh r02_pmc_psfpt --renderer psfpt > gpurun_out/r02r/pmc_psfpt.txt 2>&1
tail -12 gpurun_out/r02r/pmc_bpt.txt gpurun_out/r02r/pmc_psfpt.txt | cut -c1-250
python bench.py --renderer bpt --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_bpt.json 2> gpurun_out/r02r/b1.err
python bench.py --renderer psfpt --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_psfpt.json 2> gpurun_out/r02r/b2.err
python bench.py --renderer bpt --sc 0 --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_bpt_sc0.json 2> gpurun_out/r02r/b3.err
for f in bpt psfpt bpt_sc0; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/profiles_new/r02_bench_line_$f.json') if l.startswith('{')][-1]); r=j['roofline']
print('$f', round(j['value'],1), 'frac', round(r['frac'],3), 'traffic', r['traffic'], 'counter_frac', r['counter_frac'], r['traffic_source'][:60])"; done
