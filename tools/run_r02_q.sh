#!/bin/bash
mkdir -p gpurun_out/r02q
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_psfpt.py -m gpu -q -x 2>&1 | tail -6
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 600 python bench.py --renderer psfpt --gpus 2 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r02q/psfpt_n2_gloo.json 2> gpurun_out/r02q/psfpt_n2_gloo.err
python bench.py --renderer psfpt --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r02q/psfpt_n1.json 2> gpurun_out/r02q/psfpt_n1.err
for f in psfpt_n2_gloo psfpt_n1; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02q/$f.json') if l.startswith('{')][-1])
print('$f', round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), j['n_gpus'], j['kernel_ms_per_step'], j['config']['sharding'][:80])" || tail -5 gpurun_out/r02q/$f.err; done
