#!/bin/bash
mkdir -p gpurun_out/r02q
for b in 1 8 32 64; do
python bench.py --renderer psfpt --batch $b --steps 128 --no-cpu-baseline > gpurun_out/r02q/psf_b$b.json 2> gpurun_out/r02q/psf_b$b.err
python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02q/psf_b$b.json') if l.startswith('{')][-1])
print('psfpt batch $b', round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), j['config']['passes_in_flight'], j['kernel_ms_per_step'])" || tail -5 gpurun_out/r02q/psf_b$b.err
done
