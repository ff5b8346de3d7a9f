#!/bin/bash
mkdir -p gpurun_out/r02j
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -m gpu -x -k "batch or config2 or config3_size_standin or sharded or 4k" ) > gpurun_out/r02j/tests.log 2>&1
tail -4 gpurun_out/r02j/tests.log
for lanes in 1 2 3 4; do
FPT_PT_LANES=$lanes python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02j/bench_l${lanes}_driver.json 2> gpurun_out/r02j/bench_l${lanes}_driver.err
FPT_PT_LANES=$lanes python bench.py --no-cpu-baseline > gpurun_out/r02j/bench_l${lanes}_default.json 2> gpurun_out/r02j/bench_l${lanes}_default.err
for f in bench_l${lanes}_driver bench_l${lanes}_default; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02j/$f.json') if l.startswith('{')][-1])
print('$f', round(j['value'],1), {k:round(v,4) for k,v in j['kernel_ms_per_step'].items()}, 'frac', round(j['roofline']['frac'],3), 'chip', round(j['roofline']['frac_chip'],3))
" || tail -3 gpurun_out/r02j/$f.err; done; done
