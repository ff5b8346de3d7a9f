#!/bin/bash
mkdir -p gpurun_out/r02q
timeout 900 python -m pytest tests/test_bpt.py tests/test_gpu_full_size.py tests/test_multi_gpu.py -m gpu -q -x -k "bpt or splat" 2>&1 | tail -4
for sc in 1 0; do
python bench.py --renderer bpt --sc $sc --no-cpu-baseline > gpurun_out/r02q/bpt_sc$sc.json 2> gpurun_out/r02q/bpt_sc$sc.err
python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02q/bpt_sc$sc.json') if l.startswith('{')][-1])
print('sc $sc', round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), j['kernel_ms_per_step'])" || tail -3 gpurun_out/r02q/bpt_sc$sc.err
done
