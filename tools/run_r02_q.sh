#!/bin/bash
mkdir -p gpurun_out/r02q
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02q/n2_weak.json 2> gpurun_out/r02q/n2_weak.err
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 8 --warmup 2 --scaling strong --no-cpu-baseline > gpurun_out/r02q/n2_strong.json 2> gpurun_out/r02q/n2_strong.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02q/n1.json 2> gpurun_out/r02q/n1.err
for f in n2_weak n2_strong n1; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02q/$f.json') if l.startswith('{')][-1])
print('$f', round(j['value'],1), j['n_gpus'], j['steps'], j['scaling'], round(j['ms_per_step'],3), j['config']['passes_in_flight'], j['config'].get('passes_timed'), j['config']['sharding'][:90])" || tail -5 gpurun_out/r02q/$f.err; done
