#!/bin/bash
mkdir -p gpurun_out/r02q
for b in 64 86 93; do
  python bench.py --batch $b --steps 258 --warmup $b --no-cpu-baseline > gpurun_out/r02q/pt_b$b.json 2> gpurun_out/r02q/pt_b$b.err
  python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02q/pt_b$b.json') if l.startswith('{')][-1])
print('batch $b', round(j['value'],1), j['config']['passes_in_flight'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in j['kernel_ms_per_step'].items() if 'busy' in k})" || tail -2 gpurun_out/r02q/pt_b$b.err
done
